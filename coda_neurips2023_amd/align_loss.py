"""CLIP-space alignment losses of all decoder layers as one fused pass (csrc/align_loss.hip).

``align_loss_sums(emb, gt, wmask, text, logit_scale, labels, conf)`` returns the per-layer
UN-normalised sums of the two live alignment terms of the reference,

* ``l1[l] = sum |emb*w - gt*w|``                                (criterion.py:924-943)
* ``ce[l] = sum conf * CE(t * <emb/(|emb|+1e-32), text_j>_j, label)``  (criterion.py:598-644)

for ``emb (L,B,nq,E)``; the caller divides by the reference's normalisers.  Gradients flow to
``emb`` only (text / image embeddings and the temperature are frozen in the reference).
"""
import torch

from . import _lib


def eligible(emb, gt, text, logit_scale):
    e = emb.shape[-1]
    return (emb.is_cuda and emb.dtype == torch.float32 and emb.dim() == 4 and emb.stride(-1) == 1 and e % 64 == 0
            and e <= 1024 and not gt.requires_grad and not text.requires_grad
            and not (torch.is_tensor(logit_scale) and logit_scale.requires_grad))


class _AlignLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, gt, wmask, text, logit_scale, labels, conf):
        nl, b, nq, e = emb.shape
        ncls = text.shape[1]
        lib = _lib.load()
        partial = torch.empty((nl * b * nq, 2), dtype=torch.float32, device=emb.device)
        st = lib.coda_align_loss_fwd_f32(emb.data_ptr(), emb.stride(0), emb.stride(1), emb.stride(2), gt.data_ptr(),
                                         wmask.data_ptr(), text.data_ptr(), logit_scale.data_ptr(), labels.data_ptr(),
                                         conf.data_ptr(), nl, b, nq, e, ncls, partial.data_ptr(),
                                         _lib.current_stream_handle())
        _lib.check(st, "align_loss_fwd")
        sums = partial.view(nl, b * nq, 2).sum(1)
        ctx.save_for_backward(emb, gt, wmask, text, logit_scale, labels, conf)
        return sums[:, 0], sums[:, 1]

    @staticmethod
    def backward(ctx, g1, g2):
        emb, gt, wmask, text, logit_scale, labels, conf = ctx.saved_tensors
        nl, b, nq, e = emb.shape
        zero = torch.zeros(nl, dtype=torch.float32, device=emb.device)
        g = torch.stack([g1 if g1 is not None else zero, g2 if g2 is not None else zero], 1).contiguous()
        demb = torch.empty((nl, b, nq, e), dtype=torch.float32, device=emb.device)
        lib = _lib.load()
        st = lib.coda_align_loss_bwd_f32(emb.data_ptr(), emb.stride(0), emb.stride(1), emb.stride(2), gt.data_ptr(),
                                         wmask.data_ptr(), text.data_ptr(), logit_scale.data_ptr(), labels.data_ptr(),
                                         conf.data_ptr(), g.data_ptr(), nl, b, nq, e, text.shape[1], demb.data_ptr(),
                                         _lib.current_stream_handle())
        _lib.check(st, "align_loss_bwd")
        return demb, None, None, None, None, None, None


# From this many classes on the logits and their gradient are dense products on the matrix cores (the stage-2 prompt sets:
# 232 / 1201 classes, models/model_3detr.py:321) instead of 2 * ncls * E vector FMAs per row and direction.
GEMM_MIN_CLASSES = 64


def _shared_text(text):
    """(ncls, E) when every scene uses the same class embeddings -- an expanded view (stride 0 over the scenes, what
    model_3detr.py builds) or a single scene; None for genuinely per-scene text (the row kernel handles that)."""
    if text.shape[0] == 1 or text.stride(0) == 0:
        return text[0]
    return None


class _AlignLossGemm(torch.autograd.Function):
    """The two terms with the class logits as ONE product ehat (L*B*nq, E) . text^T and their gradient as ONE product
    dlogits . text (gemm.linear / gemm.mm: the bf16x3 matrix-core kernels for >= 8192 rows, the fp32 library below),
    row-wise pieces in csrc/align_loss.hip (coda_align_rows_*_f32, coda_align_ce_f32)."""

    @staticmethod
    def forward(ctx, emb, gt, wmask, text2, logit_scale, labels, conf):
        from . import gemm
        nl, b, nq, e = emb.shape
        rows, ncls = nl * b * nq, text2.shape[0]
        dev = emb.device
        lib = _lib.load()
        ncols = -(-ncls // 128) * 128
        tpad = torch.zeros((ncols, e), dtype=torch.float32, device=dev)
        tpad[:ncls].copy_(text2)
        gemm.declare_weight(tpad)   # split once for both products of this step (gemm.py)
        ehat = torch.empty((rows, e), dtype=torch.float32, device=dev)
        stat = torch.empty((rows, 2), dtype=torch.float32, device=dev)
        partial = torch.empty((rows, 2), dtype=torch.float32, device=dev)
        _lib.check(lib.coda_align_rows_fwd_f32(emb.data_ptr(), emb.stride(0), emb.stride(1), emb.stride(2), gt.data_ptr(),
                                               wmask.data_ptr(), nl, b, nq, e, ehat.data_ptr(), stat.data_ptr(),
                                               partial.data_ptr(), _lib.current_stream_handle()), "align_rows_fwd")
        logits = gemm.linear(ehat, tpad)                                   # (rows, ncols)
        _lib.check(lib.coda_align_ce_f32(logits.data_ptr(), logits.stride(0), ncls, ncols, logit_scale.data_ptr(),
                                         labels.data_ptr(), conf.data_ptr(), None, partial.data_ptr(), rows, b * nq,
                                         _lib.current_stream_handle()), "align_ce")
        sums = partial.view(nl, b * nq, 2).sum(1)
        ctx.save_for_backward(emb, gt, wmask, tpad, logit_scale, labels, conf, stat, logits)
        ctx.ncls = ncls
        return sums[:, 0], sums[:, 1]

    @staticmethod
    def backward(ctx, g1, g2):
        from . import gemm
        emb, gt, wmask, tpad, logit_scale, labels, conf, stat, logits = ctx.saved_tensors
        nl, b, nq, e = emb.shape
        rows, ncols = nl * b * nq, tpad.shape[0]
        dev = emb.device
        lib = _lib.load()
        zero = torch.zeros(nl, dtype=torch.float32, device=dev)
        g = torch.stack([g1 if g1 is not None else zero, g2 if g2 is not None else zero], 1).contiguous()
        # the saved logits become d loss / d logits in place (a second backward through the same graph is not supported,
        # as for every fused node of this package)
        _lib.check(lib.coda_align_ce_f32(logits.data_ptr(), logits.stride(0), ctx.ncls, ncols, logit_scale.data_ptr(),
                                         labels.data_ptr(), conf.data_ptr(), g.data_ptr(), None, rows, b * nq,
                                         _lib.current_stream_handle()), "align_ce_bwd")
        dh = gemm.mm(logits, tpad)                                          # (rows, E)
        demb = torch.empty((nl, b, nq, e), dtype=torch.float32, device=dev)
        _lib.check(lib.coda_align_rows_bwd_f32(emb.data_ptr(), emb.stride(0), emb.stride(1), emb.stride(2), gt.data_ptr(),
                                               wmask.data_ptr(), stat.data_ptr(), dh.data_ptr(), g.data_ptr(), nl, b, nq, e,
                                               demb.data_ptr(), _lib.current_stream_handle()), "align_rows_bwd")
        return demb, None, None, None, None, None, None


def align_loss_sums(emb, gt, wmask, text, logit_scale, labels, conf):
    """emb (L,B,nq,E) [any (l,b,q) strides], gt (B,nq,E), wmask (B,nq[,1]), text (B,ncls,E),
    logit_scale scalar tensor, labels (L,B,nq) int64, conf (L,B,nq) -> (l1 (L,), ce (L,))."""
    dev = emb.device
    scale = logit_scale if torch.is_tensor(logit_scale) else torch.tensor(float(logit_scale), device=dev)
    scale = scale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    text2 = _shared_text(text) if text.shape[1] >= GEMM_MIN_CLASSES else None
    if text2 is not None:
        return _AlignLossGemm.apply(emb, gt.contiguous(), wmask.reshape(wmask.shape[0], wmask.shape[1]).contiguous(),
                                    text2.to(torch.float32).contiguous(), scale, labels.contiguous(),
                                    conf.to(torch.float32).contiguous())
    return _AlignLoss.apply(emb, gt.contiguous(), wmask.reshape(wmask.shape[0], wmask.shape[1]).contiguous(),
                            text.to(torch.float32).contiguous(), scale, labels.contiguous(),
                            conf.to(torch.float32).contiguous())
