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


def align_loss_sums(emb, gt, wmask, text, logit_scale, labels, conf):
    """emb (L,B,nq,E) [any (l,b,q) strides], gt (B,nq,E), wmask (B,nq[,1]), text (B,ncls,E),
    logit_scale scalar tensor, labels (L,B,nq) int64, conf (L,B,nq) -> (l1 (L,), ce (L,))."""
    dev = emb.device
    scale = logit_scale if torch.is_tensor(logit_scale) else torch.tensor(float(logit_scale), device=dev)
    scale = scale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    return _AlignLoss.apply(emb, gt.contiguous(), wmask.reshape(wmask.shape[0], wmask.shape[1]).contiguous(),
                            text.to(torch.float32).contiguous(), scale, labels.contiguous(),
                            conf.to(torch.float32).contiguous())
