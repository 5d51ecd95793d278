"""Image side of the CLIP distillation branch: crops, embeddings, labels.

The reference's ``get_predicted_box_clip_embedding`` (models/model_3detr.py:902-1210, the stage-1 recipe) and
``get_predicted_box_clip_embedding_nms_iou_save_keep_clip_driven_with_cate_confidence`` (:1212-1632, the stage-2
recipe: ``--online_nms_update_save_novel_label_clip_driven_with_cate_confidence``) walk scenes and selected
proposals in Python: ``int(torch.min(...))`` read-backs per box, a fresh white canvas per crop, a torchvision
resize per crop, one tower call per scene, ``torchvision.ops.nms`` + a Python double loop of 3-D IoUs per scene.
Here the geometry of ALL proposals is one launch (``coda_project_box_rects_f64``: un-augment, project, clip, flip,
2-D extent, validity), the crops of all scenes one more (``coda_crop_resize_f32``: crop -> white square -> bicubic
resize -> CLIP normalisation), the frozen tower -- ``clip_tower.ImageTower`` (``coda_vit_fwd``) built from the
deployment's CLIP checkpoint, or any module with the reference's ``encode_image`` -- runs ONCE on the whole batch,
the soft-max / arg-max over the class prompts is one fused MFMA launch (``clip_labels.weak_labels``), and the
stage-2 candidate filter (2-D NMS, ground-truth overlap, objectness) one launch (``clip_labels.pseudo_box_filter``).
``RegionEmbeddingProvider`` is the ``region_embedding_provider`` of ``model_3detr.build_model``
and covers both methods, every branch: random / objectness-driven selection (:990-1004), embedding scatter + mask
(:1102-1103), novel boxes appended to the ground truth (:1107-1151), CLIP weak labels (:1153-1180, :1614-1631) and the
pseudo-label mining with its ``.npy`` files (:1292-1541).  Pinned end to end on the reference's own methods run with a
seeded stand-in tower (tests/golden/region_branch.npz, tests/test_region_branch_gpu.py).

Host read-backs: none in the default (random-selection) path; the objectness-driven selection reads one (B, K) byte
mask per step (numpy's generator must see the same candidate lists as the reference's), the mining pass -- every
``online_nms_update_save_epoch``-th epoch -- reads its results once to write the files.

Differences a caller can observe: proposals the reference skips (zero size, empty extent, behind the camera)
still occupy a slot of the tower's batch (a white image) -- their embeddings are discarded and their mask entry is
0, exactly as in the reference; equal objectness scores are ordered by proposal index in the NMS (torchvision's sort
is unstable).
"""
import numpy as np
import torch

from . import _lib, clip_labels, clip_tower


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
    """``region_embedding_provider(inputs, outputs, curr_epoch, if_test)`` for ``build_model``: adds
    ``gt_text_correlation_embedding`` (B,K,512) / ``_mask`` (B,K,1), ``weak_box_cate_label`` /
    ``weak_confidence_weight`` (B,K) to ``outputs`` -- and, in the late-epoch / stage-2 modes, novel boxes to the ground
    truth in ``inputs`` or pseudo-label rows to ``inputs["pseudo_box_path"]`` -- like the reference's two methods
    (module docstring).  Constructor keywords are the reference's flag names (main.py:37-304)."""

    def __init__(self, clip_model, distillation_box_num=32, box_pool=128, rng=None, if_select_box_by_objectness=False,
                 if_keep_box=False, keep_objectness=0.5, if_clip_weak_labels=False,
                 online_nms_update_save_novel_label_clip_driven_with_cate_confidence=False, save_objectness=0.3,
                 online_nms_update_save_epoch=50, clip_driven_keep_thres=0.3, train_range_max=10,
                 if_accumulate_former_pseudo_labels=False, objectness_epoch=540):
        self.clip_model = clip_model
        self.num = distillation_box_num
        self.pool = np.arange(box_pool, dtype=np.int8 if box_pool <= 128 else np.int64)  # (:191: int8)
        self.rng = rng if rng is not None else np.random  # the reference draws from numpy's global generator
        self.if_select_box_by_objectness = if_select_box_by_objectness
        self.if_keep_box = if_keep_box                # main.py:356 pokes model.if_keep_box: the model forwards it
        self.keep_objectness = keep_objectness
        self.if_clip_weak_labels = if_clip_weak_labels
        self.stage2 = online_nms_update_save_novel_label_clip_driven_with_cate_confidence
        self.save_objectness = save_objectness
        self.save_epoch = online_nms_update_save_epoch
        self.clip_driven_keep_thres = clip_driven_keep_thres
        self.train_range_max = train_range_max
        self.if_accumulate_former_pseudo_labels = if_accumulate_former_pseudo_labels
        self.objectness_epoch = objectness_epoch      # the literal 540 of :990, 1107

    @classmethod
    def from_args(cls, clip_model, args, rng=None):
        """The flags main.py parses (and models/model_3detr.py:436-473 copies onto the model)."""
        g = lambda k, d: getattr(args, k, d)  # noqa: E731
        return cls(clip_model, g("distillation_box_num", 32), 128, rng, g("if_select_box_by_objectness", False),
                   g("if_keep_box", False), g("keep_objectness", 0.5), g("if_clip_weak_labels", False),
                   g("online_nms_update_save_novel_label_clip_driven_with_cate_confidence", False),
                   g("save_objectness", 0.3), g("online_nms_update_save_epoch", 50), g("clip_driven_keep_thres", 0.3),
                   g("train_range_max", 10), g("if_accumulate_former_pseudo_labels", False))

    # ---- pieces ------------------------------------------------------------------------------------------------
    def _encode(self, crops):
        if isinstance(self.clip_model, clip_tower.ImageTower):  # this package's tower: float32 crops in, class embedding out
            feats = self.clip_model.encode_image(crops)
        else:
            feats = self.clip_model.encode_image(crops.to(self.clip_model.dtype) if hasattr(self.clip_model, "dtype") else crops)
        if isinstance(feats, tuple):
            feats = feats[0]
        return feats.to(torch.float32)

    def _resolution(self):
        return self.clip_model.visual.input_resolution

    def _select(self, outputs, b, k, curr_epoch):
        """Per-scene proposal lists (:989-1004 / :1547), padded to a common length: (select (B,S) int64 numpy,
        slot mask (B,S) bool numpy)."""
        pool = self.pool[self.pool < k]
        num = min(self.num, len(pool))
        if self.stage2 or (not self.if_select_box_by_objectness) or curr_epoch < self.objectness_epoch:
            lists = [self.rng.choice(pool, num, replace=False) for _ in range(b)]
        else:
            positive = (outputs["objectness_prob"].detach() > 0.05).cpu().numpy()   # the one read-back of this mode
            lists = []
            for row in positive:
                ob, bg = np.nonzero(row)[0], np.nonzero(~row)[0]
                if ob.shape[0] >= self.num:
                    lists.append(ob)
                else:
                    lists.append(np.concatenate((ob, self.rng.choice(bg, self.num - ob.shape[0], replace=False))))
        width = max(len(x) for x in lists)
        select = np.zeros((b, width), dtype=np.int64)
        slot = np.zeros((b, width), dtype=bool)
        for i, x in enumerate(lists):
            select[i, :len(x)] = x
            slot[i, :len(x)] = True
        return select, slot

    @staticmethod
    def _scatter_rows(target, pos, ok, values):
        """target[b, pos[b, s]] = values[b, s] where ok[b, s], in place, without reading anything back."""
        b, g = target.shape[:2]
        tail = target.shape[2:]
        ext = torch.cat([target, target.new_zeros((b, 1) + tuple(tail))], 1)
        idx = torch.where(ok, pos, torch.full_like(pos, g)).view(b, -1, *([1] * len(tail))).expand(b, pos.shape[1], *tail)
        ext.scatter_(1, idx, values.to(target.dtype))
        target.copy_(ext[:, :g])

    def _keep_novel_boxes(self, inputs, outputs, sel, keep, conf, label):
        """:1107-1151: selected proposals that look like an object (objectness > keep_objectness) and that CLIP assigns
        to a class beyond the first ten with probability > 0.5 become ground-truth boxes of THIS step, appended
        behind the real ones in selection order, up to slot 63."""
        present = inputs["gt_box_present"]
        g = present.shape[1]
        obj = torch.gather(outputs["objectness_prob"].detach(), 1, sel)
        novel = keep & (obj > self.keep_objectness) & (conf > 0.5) & (label > 9)
        begin = present.sum(1).long()
        pos = begin[:, None] + torch.cumsum(novel.long(), 1) - 1
        ok = novel & (pos <= min(63, g - 1))

        def pick(key):
            v = outputs[key].detach()
            return torch.gather(v, 1, sel.view(*sel.shape, *([1] * (v.dim() - 2))).expand(*sel.shape, *v.shape[2:]))

        angle_cls = pick("angle_logits").argmax(-1)                 # softmax(...).max(-1): the same index
        self._scatter_rows(present, pos, ok, torch.ones_like(obj))
        self._scatter_rows(inputs["gt_angle_class_label"], pos, ok, angle_cls)
        self._scatter_rows(inputs["gt_angle_residual_label"], pos, ok,
                           torch.gather(pick("angle_residual"), 2, angle_cls.unsqueeze(-1)).squeeze(-1))
        for dst, src in (("gt_box_sizes_normalized", "size_normalized"), ("gt_box_sizes", "size_unnormalized"),
                         ("gt_box_corners", "box_corners"), ("gt_box_corners_xyz", "box_corners_xyz"),
                         ("gt_box_angles", "angle_continuous"), ("gt_box_centers_normalized", "center_normalized"),
                         ("gt_box_centers", "center_unnormalized")):
            self._scatter_rows(inputs[dst], pos, ok, pick(src))

    def _unaugmented_boxes(self, inputs, outputs):
        """(B,K,7) float32 [centre, size, heading] in the un-augmented frame (:1239-1252, 1297-1299), in the dtype
        torch's promotion gives the reference (the dataset's augmentation arrays are float64)."""
        dev = outputs["center_unnormalized"].device
        b = outputs["center_unnormalized"].shape[0]

        def arr(key, shape):
            return torch.as_tensor(inputs[key]).to(dev).reshape(shape)

        scale, rot = arr("scale_array", (b, 1, 3)), arr("rot_array", (b, 3, 3))
        centre = outputs["center_unnormalized"].detach().clone() * scale
        size = outputs["size_unnormalized"].detach().clone() * scale
        centre = torch.matmul(centre, rot.to(centre.dtype))
        angle = outputs["angle_continuous"].detach().clone() + arr("rot_angle", (b, 1))
        if "zx_flip_array" in inputs:
            zx = arr("zx_flip_array", (b, 1))
            centre[:, :, 1] = centre[:, :, 1] * zx
            angle = torch.where(zx < 0, np.pi - angle, angle)
        flip = arr("flip_array", (b, 1))
        centre[:, :, 0] = centre[:, :, 0] * flip
        angle = torch.where(flip < 0, np.pi - angle, angle)
        return torch.cat([centre, size.to(centre.dtype), angle.unsqueeze(-1).to(centre.dtype)], -1).to(torch.float32)

    def _mine_pseudo_labels(self, inputs, outputs, rects, valid):
        """:1292-1541: candidates = NMS survivors that overlap no ground-truth box and look like objects; those CLIP
        assigns to a NOVEL class (index >= train_range_max of the full prompt list) with probability above
        ``clip_driven_keep_thres`` are written -- [centre, size, heading, class, probability, objectness] in the
        un-augmented frame -- to the scene's pseudo-label file."""
        sel, count = clip_labels.pseudo_box_filter(rects, valid, outputs["objectness_prob"], outputs["box_corners"],
                                                   inputs["gt_box_corners"], inputs["gt_box_present"], 0.25, 0.25,
                                                   self.save_objectness)
        count_h = count.cpu().numpy()                         # (this pass writes files: it synchronises anyway)
        width = int(count_h.max()) if count_h.size else 0
        if width == 0:
            return
        b = sel.shape[0]
        slots = sel[:, :width].long().clamp(min=0)
        crops = crop_resize(inputs["input_image"], slots, rects, valid, self._resolution())
        feats = self._encode(crops)
        conf, label = clip_labels.weak_labels(feats, outputs["maybe_novel_text_features_clip"], outputs["logit_scale"])
        info = self._unaugmented_boxes(inputs, outputs)
        rows = torch.cat([torch.gather(info, 1, slots.unsqueeze(-1).expand(-1, -1, 7)), label.view(b, width, 1).float(),
                          conf.view(b, width, 1),
                          torch.gather(outputs["objectness_prob"].detach().float(), 1, slots).unsqueeze(-1)], -1).cpu().numpy()
        conf_h, label_h = conf.view(b, width).cpu().numpy(), label.view(b, width).cpu().numpy()
        begin = torch.as_tensor(inputs["gt_ori_box_num"]).reshape(-1).cpu().numpy()
        for i in range(b):
            n = int(count_h[i])
            take = (conf_h[i, :n] > self.clip_driven_keep_thres) & (label_h[i, :n] >= self.train_range_max)
            if int(begin[i]) > 63 or not take.any():          # (:1520-1527: nothing is stored for a full scene)
                continue
            later = rows[i, :n][take].astype(np.float32)
            path = inputs["pseudo_box_path"][i]
            if not self.if_accumulate_former_pseudo_labels:
                np.save(path, later)
            else:
                former = np.load(path)
                np.save(path, later if former.shape[0] == 0 else np.concatenate((former, later), axis=0))

    # ---- the provider ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, inputs, outputs, curr_epoch=-1, if_test=False):
        corners = outputs["box_corners_xyz"]
        b, k = corners.shape[:2]
        dev = corners.device
        with torch.cuda.device(dev):
            rects, valid = project_box_rects(inputs, corners, outputs["size_unnormalized"])
            if self.stage2 and (not if_test) and curr_epoch % self.save_epoch == 0:
                self._mine_pseudo_labels(inputs, outputs, rects, valid)
            select, slot = self._select(outputs, b, k, curr_epoch)
            # one pinned, non-blocking host -> device copy for both arrays: a pageable copy would park the host
            # until the GPU has drained everything enqueued so far (the whole forward pass)
            staged = torch.from_numpy(np.stack((select, slot.astype(np.int64)))).pin_memory().to(dev, non_blocking=True)
            sel, slot_dev = staged[0], staged[1].bool()
            crops = crop_resize(inputs["input_image"], sel, rects, valid, self._resolution())
            feats = self._encode(crops).view(b, sel.shape[1], -1)
            keep = torch.gather(valid, 1, sel).bool() & slot_dev     # (B,S)
            keepf = keep.to(torch.float32).unsqueeze(-1)
            # padded slots (index 0, keep 0) and skipped proposals must not touch a real entry: they go to a trash row
            emb = torch.zeros((b, k + 1, feats.shape[-1]), dtype=torch.float32, device=dev)
            mask = torch.zeros((b, k + 1, 1), dtype=torch.float32, device=dev)
            dst = torch.where(keep, sel, torch.full_like(sel, k)).unsqueeze(-1)
            emb.scatter_(1, dst.expand(-1, -1, feats.shape[-1]), feats * keepf)
            mask.scatter_(1, dst, keepf)
            emb, mask = emb[:, :k].contiguous(), mask[:, :k].contiguous()
            outputs["gt_text_correlation_embedding"] = emb
            outputs["gt_text_correlation_embedding_mask"] = mask
            if (not self.stage2) and self.if_keep_box and curr_epoch >= self.objectness_epoch:
                conf, label = clip_labels.weak_labels(feats * keepf, outputs["text_features_clip"], outputs["logit_scale"])
                self._keep_novel_boxes(inputs, outputs, sel, keep, conf, label)
            if self.if_clip_weak_labels:
                conf, label = clip_labels.weak_labels(emb, outputs["text_features_clip"], outputs["logit_scale"], mask)
                outputs["weak_box_cate_label"], outputs["weak_confidence_weight"] = label, conf
            elif not self.stage2:                                  # (:1173-1180; the stage-2 method adds nothing)
                outputs["weak_box_cate_label"] = torch.zeros((b, k), dtype=torch.int64, device=dev)
                outputs["weak_confidence_weight"] = torch.zeros((b, k), dtype=torch.float32, device=dev)
        return outputs
