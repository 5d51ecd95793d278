"""Scaled-dot-product attention core of the 3DETR encoder/decoder.

``attention(q, k, v, mask, scale, dropout_p, need_weights)`` takes the
seq-first projections ``q (L,B,h,d)``, ``k/v (S,B,h,d)`` (dense or slices of a
packed in-projection) and returns ``out (L,B,h,d)`` (+ ``probs (B,h,L,S)`` when
asked).  Semantics are those of ``torch.nn.functional.multi_head_attention_forward``
as the reference uses it: softmax(q*scale @ k^T + mask) -> dropout(p) -> @ v.

The core runs in ``libcoda_hip.so`` (``include/coda_attention.h``: fused fp32-MFMA
forward and backward, the (B,h,L,S) probability tensor is never materialised).
GPU tensors only -- there is no CPU or PyTorch fallback for the core.  Only the
optional, off-hot-path ``need_weights=True`` output (head-averaged attention
maps for visualisation, transformer.py:126-137) is produced by a plain torch
softmax next to the fused output.
"""
import os

import torch

from . import _lib


_DTYPE_CODES = {"default": -1, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "bf16x3": 2, "x3": 2}


def _dtype_code(dtype):
    if isinstance(dtype, int):
        return dtype
    return _DTYPE_CODES[str(dtype).replace("torch.", "")]


def mfma_dtype(dtype):
    """'fp32', 'bf16', 'bf16x3' or 'default': operand type of the matrix-core products inside the fused attention
    kernels -- a per-call argument of coda_mha_*_opt_f32 (include/coda_attention.h; the library keeps no switch), and on
    this side a SCOPE, the only form there is (no setter: round 5):

        with attention_core.mfma_dtype("bf16"):
            loss = criterion(model(batch), batch)
        loss.backward()

    The forward passes issued inside the block use that operand type, and so do their backward passes whenever they run
    (the autograd node keeps the code it was issued with); other threads and the code outside the block are unaffected.
    Tensors stay float32 in every mode; 'bf16x3' carries each fp32 operand as three bf16 pieces and gives fp32-level
    results on the bf16 matrix cores."""
    return _lib.options(mfma_dtype=_dtype_code(dtype))


def get_mfma_dtype():
    """What a call issued now would use."""
    code = _lib.opt("mfma_dtype")
    if code < 0:
        code = _lib.load().coda_mha_get_mfma_dtype()
    return ("fp32", "bf16", "bf16x3")[code]


def merge_masks(attn_mask, key_padding_mask, bsz, h, tgt_len, src_len):
    """Boolean (B,h,L,S) mask (True = do not attend) or None."""
    mask = None
    if attn_mask is not None:
        if attn_mask.dtype != torch.bool:
            raise RuntimeError("only boolean attention masks are supported (the reference builds "
                               "boolean radius masks, models/transformer.py:154-161)")
        if attn_mask.dim() == 2:
            mask = attn_mask.view(1, 1, tgt_len, src_len).expand(bsz, h, tgt_len, src_len)
        else:
            mask = attn_mask.view(bsz, h, tgt_len, src_len)
    if key_padding_mask is not None:
        kpm = key_padding_mask.view(bsz, 1, 1, src_len).expand(bsz, h, tgt_len, src_len)
        mask = kpm if mask is None else (mask | kpm)
    return mask


def _row_layout(t):
    """(rows, B, h, d) tensor -> (tensor, ld): element (r,b,hh,c) at (r*B + b)*ld + hh*d + c.
    Accepts dense tensors and last-dim slices of a packed projection; copies otherwise."""
    rows, b, h, d = t.shape
    st = t.stride()
    if st[3] == 1 and st[2] == d and st[0] == b * st[1] and st[1] >= h * d and st[1] % 4 == 0 \
            and t.data_ptr() % 16 == 0:
        return t, st[1]
    t = t.contiguous()
    return t, h * d


def _ptr(t):
    return t.data_ptr() if t is not None else None


# Dropout seeding.  Eager mode: a fresh host seed per call from torch's CPU generator.
# Graph mode (`use_device_seed(tensor)`): the per-call host value only separates the calls of
# one step; the device word is folded in by the kernels and must be bumped between replays
# (e.g. `seed_tensor += 1` captured in the same graph), so replays draw fresh masks.
_DEVICE_SEED = None
_CALL_COUNTER = 0


def use_device_seed(seed_tensor):
    """seed_tensor: 1-element int64 CUDA tensor (or None to go back to host seeds)."""
    global _DEVICE_SEED
    _DEVICE_SEED = seed_tensor


def _next_seed():
    global _CALL_COUNTER
    if _DEVICE_SEED is None:
        return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item()), None
    _CALL_COUNTER += 1
    return (_CALL_COUNTER * 0x9E3779B97F4A7C15) & (2 ** 63 - 1), _DEVICE_SEED


class _FusedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8, scale, dropout_p):
        if not q.is_cuda:
            raise RuntimeError("CPU not supported")  # same contract as the pointnet2 ops
        if q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32:
            raise RuntimeError("attention core expects float32 tensors")
        lib = _lib.load()
        l, b, h, d = q.shape
        s = k.shape[0]
        q, ldq = _row_layout(q)
        k, ldk = _row_layout(k)
        v, ldv = _row_layout(v)
        out = torch.empty((l, b, h, d), dtype=torch.float32, device=q.device)
        lse = torch.empty((b, h, l), dtype=torch.float32, device=q.device)
        seed, seed_dev = _next_seed() if dropout_p > 0.0 else (0, None)
        with torch.cuda.device(q.device):
            dt = _lib.opt("mfma_dtype")  # this thread's option; the backward (another thread) gets it through ctx
            st = lib.coda_mha_fwd_opt_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(mask_u8), _ptr(out), _ptr(lse),
                                          b, h, l, s, d, ldq, ldk, ldv, float(scale), float(dropout_p), seed,
                                          _ptr(seed_dev), dt, _lib.current_stream_handle())
        _lib.check(st, "mha_fwd")
        ctx.save_for_backward(q, k, v, mask_u8, out, lse)
        ctx.meta = (ldq, ldk, ldv, float(scale), float(dropout_p), seed, seed_dev, dt)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, mask_u8, out, lse = ctx.saved_tensors
        ldq, ldk, ldv, scale, dropout_p, seed, seed_dev, dt = ctx.meta
        lib = _lib.load()
        l, b, h, d = q.shape
        s = k.shape[0]
        dout = dout.contiguous()
        dq = torch.empty((l, b, h, d), dtype=torch.float32, device=q.device)
        dk = torch.empty((s, b, h, d), dtype=torch.float32, device=q.device)
        dv = torch.empty((s, b, h, d), dtype=torch.float32, device=q.device)
        delta = torch.empty((b, h, l), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            ws, ws_bytes = backward_workspace(b, h, l, s, d, q.device, dt) if mask_u8 is None else (None, 0)
            st = lib.coda_mha_bwd_ws_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(mask_u8), _ptr(out), _ptr(lse),
                                         _ptr(dout), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(delta), b, h, l, s, d,
                                         ldq, ldk, ldv, 0, 0, 0, scale, dropout_p, seed, _ptr(seed_dev), _ptr(ws),
                                         ws_bytes, dt, _lib.current_stream_handle())
        _lib.check(st, "mha_bwd")
        return dq, dk, dv, None, None, None


def attention(q, k, v, mask, scale, dropout_p, need_weights):
    mask_u8 = None
    if mask is not None:
        mask_u8 = mask.contiguous().to(torch.uint8)
    d = q.shape[-1]
    if d in (64, 128):
        out = _FusedAttention.apply(q, k, v, mask_u8, scale, dropout_p)
    elif d < 128:
        # the kernels are specialised for the head dims of the model (64: dec_dim 256, 128:
        # dec_dim 512); other widths are zero-padded, which changes neither scores nor outputs
        pad = (64 if d < 64 else 128) - d
        qp, kp, vp = (torch.nn.functional.pad(t, (0, pad)) for t in (q, k, v))
        out = _FusedAttention.apply(qp, kp, vp, mask_u8, scale, dropout_p)[..., :d]
    else:
        raise RuntimeError(f"head_dim {d} > 128 is not supported by the fused attention core")
    probs = None
    if need_weights:
        with torch.no_grad():  # visualisation output only; not part of the differentiated path
            scores = torch.einsum("lbhd,sbhd->bhls", q * scale, k)
            if mask is not None:
                scores = scores.masked_fill(mask, float("-inf"))
            probs = torch.softmax(scores, dim=-1)
    return out, probs


# ---- measurement aid (bench.py): per-kernel HIP-event timing inside the C library ----------------
# dqg: dQ as the dS K GEMM (coda_mha_bwd_ws_f32); bwdf: dK, dV and partial dQ tiles in one kernel; dqr: their sum;
# ktp: the bf16 K^T pieces of the bf16x3 dS K GEMM
TIMING_KINDS = ("fwd", "delta", "dkv", "dq", "dqg", "bwdf", "dqr", "ktp")


def backward_workspace(b, h, l, s, d, dev, dt):
    """(tensor or None, bytes): the dS workspace of coda_mha_bwd_ws_f32 for this problem -- long unmasked sequences at
    head width 64 under fp32 MFMA operands (the encoder's self-attention); scratch from torch's caching allocator."""
    lib = _lib.load()
    if dt > 0 or (dt < 0 and lib.coda_mha_get_mfma_dtype() != 0) or os.environ.get("CODA_ATTN_DS", "1") == "0":
        return None, 0
    nbytes = lib.coda_mha_bwd_ws_bytes(b, h, l, s, d)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes // 4, dtype=torch.float32, device=dev), nbytes


def enable_kernel_timing(min_len=0, kinds=None):
    """Bracket every attention kernel of problems with l, s >= min_len with HIP events on its launch
    stream (coda_mha_timing_enable); drops earlier records.  kinds: names from TIMING_KINDS to record (default all)."""
    if kinds is None:
        _lib.check(_lib.load().coda_mha_timing_enable(int(min_len)), "coda_mha_timing_enable")
        return
    mask = 0
    for k in kinds:
        mask |= 1 << TIMING_KINDS.index(k)
    _lib.check(_lib.load().coda_mha_timing_enable_kinds(int(min_len), mask), "coda_mha_timing_enable_kinds")


def disable_kernel_timing():
    _lib.check(_lib.load().coda_mha_timing_enable(-1), "coda_mha_timing_enable")


def collect_kernel_timing(cap=16384):
    """{(kind, l, s): [ms, ...]} of the launches recorded since enable_kernel_timing()."""
    import ctypes
    kind, l, s = ((ctypes.c_int * cap)() for _ in range(3))
    ms = (ctypes.c_float * cap)()
    n = _lib.load().coda_mha_timing_collect(kind, l, s, ms, cap)
    if n < 0:
        raise RuntimeError(f"coda_mha_timing_collect failed ({n})")
    out = {}
    for i in range(n):
        out.setdefault((TIMING_KINDS[kind[i]], l[i], s[i]), []).append(ms[i])
    return out
