"""Labels from CLIP image embeddings (include/coda_clip_labels.h): the weak labels of the alignment loss and the
stage-2 pseudo-label candidate filter.  Host side of ``csrc/pseudo_label.hip``; reference lines in each docstring."""
import torch

from . import _lib


def weak_labels(embedding, text, logit_scale, mask=None):
    """``softmax(normalise(embedding) @ text^T * logit_scale).max(-1)`` without materialising any of it
    (models/model_3detr.py:1153-1172 / 1614-1631; with a 2-D ``text`` and no mask: the novel-box classifier of
    :1110-1123 / 1497-1505).

    embedding (..., 512) float32; text (ncls, 512) -- one prompt set for every row -- or (B, ncls, 512) with
    embedding (B, K, 512); logit_scale a 0-dim tensor (or a number); mask (..., 1) / (...) float32 or None: rows
    with mask < 1 get confidence 0.  -> (confidence (...) float32, label (...) int64)."""
    emb = embedding.detach()
    if not emb.is_cuda:
        raise RuntimeError("CPU not supported")
    if emb.dtype != torch.float32 or emb.shape[-1] != 512:
        raise RuntimeError("weak_labels expects float32 embeddings of width 512")
    lead = emb.shape[:-1]
    emb = emb.reshape(-1, 512)
    if emb.stride(-1) != 1 or emb.stride(0) % 4 or emb.data_ptr() % 16:
        emb = emb.contiguous()
    rows = emb.shape[0]
    text = text.detach().to(device=emb.device, dtype=torch.float32).contiguous()
    if text.dim() == 2:
        nsets, rows_per_set = 1, max(rows, 1)
    else:
        nsets = text.shape[0]
        if len(lead) != 2 or lead[0] != nsets:
            raise RuntimeError("weak_labels: text (B, ncls, 512) needs embeddings (B, K, 512)")
        rows_per_set = lead[1]
    ncls = text.shape[-2]
    scale = torch.as_tensor(logit_scale, dtype=torch.float32, device=emb.device).reshape(1)
    rm = None
    if mask is not None:
        rm = mask.detach().to(torch.float32).reshape(-1).contiguous()
        assert rm.numel() == rows
    score = torch.empty(rows, dtype=torch.float32, device=emb.device)
    label = torch.empty(rows, dtype=torch.int64, device=emb.device)
    lib = _lib.load()

    def launch(e, t, m, s, lab, n, per_set, sets):
        st = lib.coda_clip_weak_labels_f32(e.data_ptr(), e.stride(0), t.data_ptr(), scale.data_ptr(),
                                           m.data_ptr() if m is not None else None, s.data_ptr(), lab.data_ptr(), n,
                                           per_set, sets, ncls, 512, _lib.current_stream_handle())
        _lib.check(st, "coda_clip_weak_labels_f32")

    with torch.cuda.device(emb.device):
        if nsets > 1 and rows_per_set % 32:  # a 32-row tile must not straddle two prompt sets: one launch per set
            for b in range(nsets):
                sl = slice(b * rows_per_set, (b + 1) * rows_per_set)
                launch(emb[sl], text[b], None if rm is None else rm[sl], score[sl], label[sl], rows_per_set, rows_per_set, 1)
        elif rows:
            launch(emb, text, rm, score, label, rows, rows_per_set, nsets)
    return score.view(lead), label.view(lead)


def pseudo_box_filter(rects, valid, objectness, pred_corners, gt_corners, gt_present, nms_iou=0.25, gt_iou=0.25,
                      min_objectness=0.3):
    """Stage-2 pseudo-label candidates (models/model_3detr.py:1305-1426): 2-D NMS of the projected rectangles in
    objectness order, minus proposals overlapping a ground-truth box in 3-D, minus low objectness.
    -> (sel (B,K) int32: proposal indices in NMS order, -1 padded; count (B) int32)."""
    if not rects.is_cuda:
        raise RuntimeError("CPU not supported")
    b, k = valid.shape
    g = gt_present.shape[1]
    dev = rects.device
    f32 = dict(device=dev, dtype=torch.float32)
    sel = torch.empty((b, k), dtype=torch.int32, device=dev)
    count = torch.empty((b,), dtype=torch.int32, device=dev)
    args = [rects.to(torch.int32).contiguous(), valid.to(torch.uint8).contiguous(), objectness.detach().to(**f32).contiguous(),
            pred_corners.detach().to(**f32).contiguous(), gt_corners.detach().to(**f32).contiguous(),
            gt_present.detach().to(**f32).contiguous()]
    with torch.cuda.device(dev):
        st = _lib.load().coda_pseudo_box_filter_f32(*[a.data_ptr() for a in args], float(nms_iou), float(gt_iou),
                                                    float(min_objectness), sel.data_ptr(), count.data_ptr(), b, k, g,
                                                    _lib.current_stream_handle())
    _lib.check(st, "coda_pseudo_box_filter_f32")
    return sel, count
