"""``linear(x, w, b)``: ``F.linear`` whose weight gradient is a split-K GEMM.

Forward and input gradient are the plain library GEMMs.  The weight gradient
``dW = dY^T X`` of the token-wise layers of the path reduces over 16 384 tokens (encoder, the
decoder's memory projections) or 10^6 grouped rows (set-abstraction MLP) into a 256x256-ish
matrix; rocBLAS runs that as a handful of output tiles with one long K loop (11 TFLOP/s
measured).  Splitting the reduction into row chunks turns it into a batched GEMM that fills
the chip plus a small sum (3.5x - 4x faster, tools/bench_tn.py); the result differs from the
single GEMM only in fp32 summation order.
"""
import torch
import torch.nn.functional as F

from . import gemm

import os

_MIN_ROWS = 4096
_OWN_CHUNK_SUM = os.environ.get("CODA_CHUNK_SUM", "1") != "0"  # 0: torch's sum over the chunks (A/B)
_CHUNK = 2048


def _chunk_sum(part):
    """part (chunks, Co, Ci) -> sum over the chunks, (Co, Ci): the own fixed-order reduction (four columns per thread
    for up to 16 chunks, csrc/token_ln.hip colsum_finalize_grouped_kernel) instead of torch's sum over dim 0 -- 32.8 us
    for the 8 x 2048 x 256 partials of the decoder's memory projections, 17 MB."""
    if not (_OWN_CHUNK_SUM and part.is_cuda and part.dtype == torch.float32 and part.is_contiguous() and gemm.GROUPED_TN) \
            or part.shape[0] > 16:
        return part.sum(0)
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    d = gemm.DeferredWeightGrads(sums_only=True)
    d.add_colsum(part, out, part.shape[0], out.numel())
    d.flush()
    return out


def tn_gemm(dy, x):
    """dy (P, Co), x (P, Ci) -> dy^T x (Co, Ci), reduction over P split into chunks."""
    p = dy.shape[0]
    # (row chunks of the ~10^6-row SA products as one grouped launch of the own kernel + a sum measure 505 vs 630 us
    # stand-alone, tools/bench_long_tn.py, but 958 vs 979 scenes/s in the SA-only step -- 32 descriptor rows and
    # slices on the host per call -- so the batched library GEMM stays)
    if p >= _MIN_ROWS:
        part = gemm.x3_tn_partials(dy, x) if p < (1 << 18) else None
        if part is not None:
            return _chunk_sum(part)
        rows = _CHUNK if p % _CHUNK == 0 else 0
        if p >= (1 << 18) and p % 16384 == 0:
            rows = 16384
        if rows and dy.stride(1) == 1 and x.stride(1) == 1:  # (padded row strides are fine: the chunks are views)
            nc = p // rows
            return _chunk_sum(torch.bmm(dy.unflatten(0, (nc, rows)).transpose(1, 2), x.unflatten(0, (nc, rows))))
    return gemm.mm_tn(dy, x)  # (falls back to torch.mm for anything but 2-D fp32 CUDA operands)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = tn_gemm(dy2.contiguous(), x.reshape(-1, x.shape[-1]).contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return dx, dw, db


def linear(x, w, b=None):
    rows = x.numel() // x.shape[-1]
    if rows < _MIN_ROWS or not (x.requires_grad or w.requires_grad) or not torch.is_grad_enabled():
        return F.linear(x, w, b)
    return _Linear.apply(x, w, b)
