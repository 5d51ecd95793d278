"""Seeded inputs of the image-branch fixtures (tests/golden/region_branch.npz), shared by the generator -- which
feeds them to the REFERENCE's ``get_predicted_box_clip_embedding*`` methods -- and by tests/test_region_branch_gpu.py,
which feeds them to this package's ``RegionEmbeddingProvider``.  numpy ``RandomState`` and integer arithmetic only,
so both sides build bit-identical arrays; the fixture itself carries outputs only."""
import numpy as np
import torch

B, K, G, NTEXT, NSEEN = 3, 128, 64, 46, 10
IMAGE_HW = (531, 730)

CASES = {  # name: (reference method, curr_epoch, flags of the model object)
    "stage1_random": ("get_predicted_box_clip_embedding", 10,
                      dict(if_select_box_by_objectness=False, if_keep_box=False, if_clip_weak_labels=False)),
    "stage1_objectness_keep_weak": ("get_predicted_box_clip_embedding", 600,
                                    dict(if_select_box_by_objectness=True, if_keep_box=True, if_clip_weak_labels=True)),
    "stage2_mining": ("get_predicted_box_clip_embedding_nms_iou_save_keep_clip_driven_with_cate_confidence", 50,
                      dict(if_clip_weak_labels=True, if_accumulate_former_pseudo_labels=False)),
    "stage2_mining_accumulate": ("get_predicted_box_clip_embedding_nms_iou_save_keep_clip_driven_with_cate_confidence",
                                 100, dict(if_clip_weak_labels=True, if_accumulate_former_pseudo_labels=True)),
    "stage2_plain": ("get_predicted_box_clip_embedding_nms_iou_save_keep_clip_driven_with_cate_confidence", 7,
                     dict(if_clip_weak_labels=False)),
}
GT_KEYS = ["gt_box_present", "gt_angle_class_label", "gt_angle_residual_label", "gt_box_sizes_normalized",
           "gt_box_sizes", "gt_box_corners", "gt_box_corners_xyz", "gt_box_angles", "gt_box_centers_normalized",
           "gt_box_centers"]
MODEL_FLAGS = dict(distillation_box_num=32, keep_objectness=0.5, save_objectness=0.3, online_nms_update_save_epoch=50,
                   clip_driven_keep_thres=0.3, train_range_max=NSEEN)


def images():
    """(B,H,W,3) uint8, smooth + structured, from integer arithmetic (identical on every platform)."""
    h, w = IMAGE_HW
    yy, xx = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    out = np.empty((B, h, w, 3), dtype=np.uint8)
    for b in range(B):
        for c in range(3):
            p = [3 + 2 * b + c, 5 + b + 3 * c, 7 + c, 11 + 5 * b]
            v = (xx * p[0] + yy * p[1] + ((xx * yy) >> 7) * p[2] + ((xx >> 5) * (yy >> 5) * p[3] * 9)) >> 2
            out[b, :, :, c] = (v & 255).astype(np.uint8)
    return out


def build(corners_xyz_fn, corners_fn, seed=31):
    """-> (inputs, outputs, tower_weight): dictionaries of CPU torch tensors as the dataset / ``get_box_predictions``
    produce them.  ``corners_xyz_fn(size, angle, centre)`` / ``corners_fn(size, angle, centre_camera)`` are the
    corner builders (utils/box_util.py:383-490; the reference's in the generator, this package's in the test)."""
    rs = np.random.RandomState(seed)
    f64, f32 = np.float64, np.float32
    t = torch.from_numpy

    def cam_flip(c):  # depth -> camera frame (utils/box_util / flip_axis_to_camera)
        return np.stack((c[..., 0], -c[..., 2], c[..., 1]), -1)

    kmat = np.zeros((B, 3, 3), f64)
    kmat[:, 0, 0] = 520 + 20 * rs.rand(B)
    kmat[:, 1, 1] = 520 + 20 * rs.rand(B)
    kmat[:, 0, 2] = 320 + 10 * rs.rand(B)
    kmat[:, 1, 2] = 240 + 10 * rs.rand(B)
    kmat[:, 2, 2] = 1
    tilt = (rs.rand(B) - 0.5) * 0.3
    rtilt = np.zeros((B, 3, 3), f64)
    rtilt[:, 0, 0] = 1
    rtilt[:, 1, 1], rtilt[:, 1, 2] = np.cos(tilt), -np.sin(tilt)
    rtilt[:, 2, 1], rtilt[:, 2, 2] = np.sin(tilt), np.cos(tilt)
    ang = (rs.rand(B) - 0.5) * 0.6
    rot = np.zeros((B, 3, 3), f64)
    rot[:, 0, 0], rot[:, 0, 1] = np.cos(ang), -np.sin(ang)
    rot[:, 1, 0], rot[:, 1, 1] = np.sin(ang), np.cos(ang)
    rot[:, 2, 2] = 1
    flip = np.where(rs.rand(B, 1) < 0.5, -1.0, 1.0).astype(f64)
    flip[0, 0], flip[1, 0] = 1.0, -1.0
    h, w = IMAGE_HW
    ow, oh = np.full(B, 640.0), np.full(B, 480.0)
    inputs = {"K": kmat, "Rtilt": rtilt, "rot_array": rot, "rot_angle": ang, "scale_array": 0.9 + 0.2 * rs.rand(B, 1, 3),
              "flip_array": flip, "image_flip_array": np.where(flip < 0, 0.0, 1.0), "flip_length": np.full(B, float(w)),
              "ori_width": ow, "ori_height": oh, "y_offset": (w - ow) // 2, "x_offset": (h - oh) // 2}
    inputs = {k: t(np.ascontiguousarray(v, dtype=f64)) for k, v in inputs.items()}
    inputs["input_image"] = t(images())

    centres = np.stack(((rs.rand(B, K) - 0.5) * 6, 1.0 + rs.rand(B, K) * 5, (rs.rand(B, K) - 0.5) * 2), -1).astype(f32)
    centres[:, :3, 1] = -2.0                                      # behind the camera
    sizes = (rs.rand(B, K, 3) * 1.5 + 0.2).astype(f32)
    sizes[0, 5] = 0.0                                             # a zero-size proposal
    angles = ((rs.rand(B, K) - 0.5) * 3).astype(f32)
    objectness = np.stack((rs.rand(K) ** 0.5, rs.rand(K) ** 12, rs.rand(K) ** 2)).astype(f32)  # many / few / some positives
    extent = np.array([8.0, 8.0, 3.0], f32)
    mn = np.array([-4.0, 0.0, -1.5], f32)
    angle_logits = rs.randn(B, K, 12).astype(f32)
    angle_residual = ((rs.rand(B, K, 12) - 0.5) * 0.2).astype(f32)
    tc, ts, ta = t(centres), t(sizes), t(angles)
    outputs = {"box_corners_xyz": corners_xyz_fn(ts, ta, tc), "box_corners": corners_fn(ts, ta, t(cam_flip(centres))),
               "objectness_prob": t(objectness), "size_unnormalized": ts, "size_normalized": t(sizes / extent),
               "center_unnormalized": tc, "center_normalized": t((centres - mn) / extent), "angle_continuous": ta,
               "angle_logits": t(angle_logits), "angle_residual": t(angle_residual)}

    text = rs.randn(NTEXT, 512)
    text = (text / np.linalg.norm(text, axis=1, keepdims=True)).astype(f32)
    outputs["text_features_all"] = t(text)                        # (46,512): the prompts of every class
    outputs["logit_scale"] = torch.tensor(100.0)

    # ground truth: 5 / 0 / 62 boxes; some coincide with proposals (3-D IoU > 0.25 against them)
    nact = np.array([5, 0, 62])
    gsz = (rs.rand(B, G, 3) * 1.2 + 0.3).astype(f32)
    gct = np.stack(((rs.rand(B, G) - 0.5) * 6, 1.0 + rs.rand(B, G) * 5, (rs.rand(B, G) - 0.5) * 2), -1).astype(f32)
    gan = ((rs.rand(B, G) - 0.5) * 3).astype(f32)
    for b in range(B):
        for j in range(0, min(nact[b], 12), 2):                   # every other GT box sits on proposal 20 + 3j
            src = 20 + 3 * j
            gsz[b, j], gct[b, j], gan[b, j] = sizes[b, src] * 1.05, centres[b, src] + 0.02, angles[b, src]
    acls = rs.randint(0, 12, (B, G))
    tg, tgs, tga = t(gct), t(gsz), t(gan)
    inputs.update({"gt_box_present": t((np.arange(G)[None] < nact[:, None]).astype(f32)),
                   "gt_ori_box_num": t(nact.astype(np.int64)),
                   "gt_angle_class_label": t(acls.astype(np.int64)),
                   "gt_angle_residual_label": t(((rs.rand(B, G) - 0.5) * 0.2).astype(f32)),
                   "gt_box_sizes_normalized": t(gsz / extent), "gt_box_sizes": tgs,
                   "gt_box_corners": corners_fn(tgs, tga, t(cam_flip(gct))), "gt_box_corners_xyz": corners_xyz_fn(tgs, tga, tg),
                   "gt_box_angles": tga, "gt_box_centers_normalized": t((gct - mn) / extent), "gt_box_centers": tg})
    tower_w = t((rs.randn(48, 512) / 7.0).astype(f32))
    return inputs, outputs, tower_w


class StandInTower(torch.nn.Module):
    """A frozen stand-in for CLIP's image tower with the reference's ``encode_image`` interface: 4 x 4 average pooling
    of the normalised crop, then a seeded projection to 512 -- deterministic, crop-sensitive, cheap."""

    def __init__(self, weight):
        super().__init__()
        self.register_buffer("weight", weight)
        self.visual = type("V", (), {"input_resolution": 224})()

    def encode_image(self, x):
        return torch.nn.functional.adaptive_avg_pool2d(x.float(), 4).flatten(1) @ self.weight
