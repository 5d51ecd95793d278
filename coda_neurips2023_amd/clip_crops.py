"""Image side of the CLIP distillation branch up to the tower's input.

The reference's ``get_predicted_box_clip_embedding`` (models/model_3detr.py:902-1086) walks scenes and selected
proposals in Python: ``int(torch.min(...))`` read-backs per box, a fresh white canvas per crop, a torchvision
resize per crop, one tower call per scene.  Here the geometry of ALL proposals is one launch
(``coda_project_box_rects_f64``: un-augment, project, clip, flip, 2-D extent, validity), the crops of all scenes
one more (``coda_crop_resize_f32``: crop -> white square -> bicubic resize -> CLIP normalisation), and the frozen
tower -- ``clip_tower.ImageTower`` (``coda_vit_fwd``) built from the deployment's CLIP checkpoint, or any module
with the reference's ``encode_image`` -- runs ONCE on the whole batch.
Nothing reads device memory back.  ``RegionEmbeddingProvider`` is a drop-in ``region_embedding_provider`` for
``model_3detr.build_model``.

Differences a caller can observe: proposals the reference skips (zero size, empty extent, behind the camera)
still occupy a slot of the tower's batch (a white image) -- their embeddings are discarded and their mask entry is
0, exactly as in the reference; the selection of proposals is the reference's first branch
(``np.random.choice`` of 32 of the first 128 proposals, :990-991; the objectness-driven selection and the novel-box
discovery of late epochs, :992-1005 and :1087-1210, stay with the reference's code).
"""
import numpy as np
import torch

from . import _lib, clip_tower


def _f64(t, dev, shape):
    return torch.as_tensor(t).to(device=dev, dtype=torch.float64).reshape(shape).contiguous()


def project_box_rects(inputs, box_corners_xyz, size_unnormalized, want_uv=False):
    """(B,K,8,3) predicted corners in the augmented depth frame -> rects (B,K,4) int32 [xmin,ymin,xmax,ymax] in the
    padded image, valid (B,K) uint8 [, uv (B,K,8,2), depth (B,K,8) float64]."""
    corners = box_corners_xyz.detach().to(torch.float32).contiguous()
    if not corners.is_cuda:
        raise RuntimeError("CPU not supported")
    dev = corners.device
    b, k = corners.shape[:2]
    sizes = size_unnormalized.detach().to(torch.float32).contiguous()
    scale = _f64(inputs["scale_array"], dev, (b, 3))
    rot = _f64(inputs["rot_array"], dev, (b, 3, 3))
    flip = _f64(inputs["flip_array"], dev, (b,))
    zx = _f64(inputs["zx_flip_array"], dev, (b,)) if "zx_flip_array" in inputs else None
    kmat = _f64(inputs["K"], dev, (b, 3, 3))
    rtilt = _f64(inputs["Rtilt"], dev, (b, 3, 3))
    ori_wh = torch.stack((_f64(inputs["ori_width"], dev, (b,)), _f64(inputs["ori_height"], dev, (b,))), 1).contiguous()
    off = torch.stack((_f64(inputs["y_offset"], dev, (b,)), _f64(inputs["x_offset"], dev, (b,))), 1).contiguous()
    iflip = _f64(inputs["image_flip_array"], dev, (b,))
    flen = _f64(inputs["flip_length"], dev, (b,))
    rects = torch.empty((b, k, 4), dtype=torch.int32, device=dev)
    valid = torch.empty((b, k), dtype=torch.uint8, device=dev)
    uv = torch.empty((b, k, 8, 2), dtype=torch.float64, device=dev) if want_uv else None
    depth = torch.empty((b, k, 8), dtype=torch.float64, device=dev) if want_uv else None
    with torch.cuda.device(dev):
        st = _lib.load().coda_project_box_rects_f64(
            corners.data_ptr(), sizes.data_ptr(), scale.data_ptr(), rot.data_ptr(), flip.data_ptr(),
            zx.data_ptr() if zx is not None else None, kmat.data_ptr(), rtilt.data_ptr(), ori_wh.data_ptr(),
            off.data_ptr(), iflip.data_ptr(), flen.data_ptr(), uv.data_ptr() if want_uv else None,
            depth.data_ptr() if want_uv else None, rects.data_ptr(), valid.data_ptr(), b, k, _lib.current_stream_handle())
    _lib.check(st, "coda_project_box_rects_f64")
    return (rects, valid, uv, depth) if want_uv else (rects, valid)


def crop_resize(input_image, select, rects, valid, resolution=224):
    """input_image (B,H,W,3) uint8 RGB, select (B,S) proposal indices -> (B*S,3,res,res) float32, CLIP-normalised."""
    img = input_image.contiguous()
    if not img.is_cuda or img.dtype != torch.uint8:
        raise RuntimeError("crop_resize expects a uint8 CUDA image batch (B,H,W,3)")
    b, h, w, _ = img.shape
    sel = select.to(device=img.device, dtype=torch.int32).contiguous()
    s = sel.shape[1]
    out = torch.empty((b * s, 3, resolution, resolution), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        st = _lib.load().coda_crop_resize_f32(img.data_ptr(), sel.data_ptr(), rects.data_ptr(), valid.data_ptr(),
                                              out.data_ptr(), b, h, w, rects.shape[1], s, resolution,
                                              _lib.current_stream_handle())
    _lib.check(st, "coda_crop_resize_f32")
    return out


class RegionEmbeddingProvider:
    """``region_embedding_provider(inputs, outputs, curr_epoch)`` for ``build_model``: fills
    ``gt_text_correlation_embedding`` (B,K,512) / ``_mask`` (B,K,1) from the CLIP image tower on the predicted boxes'
    crops (models/model_3detr.py:902-1086, first selection branch)."""

    def __init__(self, clip_model, distillation_box_num=32, box_pool=128, rng=None):
        self.clip_model = clip_model
        self.num = distillation_box_num
        self.pool = np.arange(box_pool)
        self.rng = rng if rng is not None else np.random  # the reference draws from numpy's global generator

    @torch.no_grad()
    def __call__(self, inputs, outputs, curr_epoch=-1):
        corners = outputs["box_corners_xyz"]
        b, k = corners.shape[:2]
        dev = corners.device
        rects, valid = project_box_rects(inputs, corners, outputs["size_unnormalized"])
        pool = self.pool[self.pool < k]
        select = np.stack([self.rng.choice(pool, min(self.num, len(pool)), replace=False) for _ in range(b)])
        sel = torch.from_numpy(select.astype(np.int64)).to(dev)
        crops = crop_resize(inputs["input_image"], sel, rects, valid, self.clip_model.visual.input_resolution)
        if isinstance(self.clip_model, clip_tower.ImageTower):  # this package's tower: float32 crops in, class embedding out
            feats = self.clip_model.encode_image(crops)
        else:
            feats = self.clip_model.encode_image(crops.to(self.clip_model.dtype) if hasattr(self.clip_model, "dtype") else crops)
        if isinstance(feats, tuple):
            feats = feats[0]
        feats = feats.to(torch.float32).view(b, sel.shape[1], -1)
        keep = torch.gather(valid, 1, sel).to(torch.float32).unsqueeze(-1)   # (B,S,1)
        emb = torch.zeros((b, k, feats.shape[-1]), dtype=torch.float32, device=dev)
        mask = torch.zeros((b, k, 1), dtype=torch.float32, device=dev)
        emb.scatter_(1, sel.unsqueeze(-1).expand(-1, -1, feats.shape[-1]), feats * keep)
        mask.scatter_(1, sel.unsqueeze(-1), keep)
        outputs["gt_text_correlation_embedding"] = emb
        outputs["gt_text_correlation_embedding_mask"] = mask
        return outputs
