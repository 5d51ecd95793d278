"""CPU restatement of the image side of the CLIP distillation branch -- TEST INFRASTRUCTURE ONLY.

Follows models/model_3detr.py:907-965 (un-augment, project, clip, offsets, flip), datasets/sunrgbd_utils.py:
602-635 (projection), models/model_3detr.py:1011-1066 (per-box extent, validity, white-square padding) and
CLIP/clip/clip.py:95-101 (the tensor transform), numpy float64 / torch-CPU.  The projection function is pinned on
vectors produced by the reference's own ``project_3dpoint_to_2dpoint_corners_tensor`` (tests/golden/
clip_crops.npz).  The resize is ``torch.nn.functional.interpolate(mode="bicubic", align_corners=False)`` -- the
function torchvision 0.9.1's ``Resize`` calls for tensors (README.md:43 names that version; torchvision itself is
not in this image, so its thin wrapper -- clamp, round, cast for uint8 -- is restated here: parity with the wrapper
is unpinned).  Only tests/ may import this."""
import numpy as np
import torch


def project(corners_xyz, inp):
    """(B,K,8,3) -> uv (B,K,8,2), depth (B,K,8) float64."""
    g = np.asarray(corners_xyz, np.float32).astype(np.float64) * np.asarray(inp["scale_array"], np.float64).reshape(-1, 1, 1, 3)
    g = np.matmul(g, np.asarray(inp["rot_array"], np.float64)[:, None])
    if "zx_flip_array" in inp:
        g[..., 1] = g[..., 1] * np.asarray(inp["zx_flip_array"], np.float64).reshape(-1, 1, 1)
    g[..., 0] = g[..., 0] * np.asarray(inp["flip_array"], np.float64).reshape(-1, 1, 1)
    rt = np.asarray(inp["Rtilt"], np.float64)[:, None]                      # (B,1,3,3)
    pc2 = np.matmul(np.swapaxes(rt, 2, 3), np.swapaxes(g, 2, 3))           # (B,K,3,8)
    pc2 = np.swapaxes(pc2, 2, 3)
    cam = np.stack((pc2[..., 0], -pc2[..., 2], pc2[..., 1]), -1)
    uvw = np.matmul(cam, np.swapaxes(np.asarray(inp["K"], np.float64)[:, None], 2, 3))
    uv = np.stack((uvw[..., 0] / (uvw[..., 2] + 1e-32), uvw[..., 1] / (uvw[..., 2] + 1e-32)), -1)
    return uv, uvw[..., 2]


def rects(corners_xyz, sizes, inp):
    """-> uv after clipping / offsets / flip, depth, rects (B,K,4) int32, valid (B,K) bool."""
    uv, depth = project(corners_xyz, inp)
    b = uv.shape[0]
    ow = np.asarray(inp["ori_width"], np.float64).reshape(b, 1, 1)
    oh = np.asarray(inp["ori_height"], np.float64).reshape(b, 1, 1)
    u = np.clip(uv[..., 0], 0, ow - 1) + np.asarray(inp["y_offset"], np.float64).reshape(b, 1, 1)
    v = np.clip(uv[..., 1], 0, oh - 1) + np.asarray(inp["x_offset"], np.float64).reshape(b, 1, 1)
    ifl = np.asarray(inp["image_flip_array"], np.float64).reshape(b, 1, 1)
    fl = np.asarray(inp["flip_length"], np.float64).reshape(b, 1, 1)
    u = u * ifl + (1 - ifl) * (fl - 1 - u)
    r = np.stack((u.min(-1), v.min(-1), u.max(-1), v.max(-1)), -1).astype(np.int32)   # int(): truncation
    smax = np.asarray(sizes, np.float32).max(-1)
    valid = ~(smax < 1e-16) & (r[..., 2] - r[..., 0] > 0) & (r[..., 3] - r[..., 1] > 0) & ~(depth.min(-1) < 0)
    return np.stack((u, v), -1), depth, r, valid


MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def crop_resize(image, rect, res=224):
    """image (H,W,3) uint8 tensor, rect [xmin,ymin,xmax,ymax] -> (3,res,res) float32, CLIP-normalised (:1027-1066 and
    CLIP/clip/clip.py:95-101)."""
    xmin, ymin, xmax, ymax = (int(x) for x in rect)
    crop = image[ymin:ymax, xmin:xmax]
    w, h = crop.shape[0], crop.shape[1]           # (sic) the reference's names: w = rows, h = columns
    edge = max(w, h)
    canvas = torch.ones(edge, edge, 3, dtype=torch.uint8) * 255
    yb, xb = (edge - w) // 2, (edge - h) // 2
    canvas[yb:yb + w, xb:xb + h] = crop
    x = canvas.permute(2, 0, 1).unsqueeze(0).float()
    y = torch.nn.functional.interpolate(x, size=(res, res), mode="bicubic", align_corners=False)
    y = y.clamp(0, 255).round()[0]                # torchvision 0.9.1 resize on uint8 tensors: clamp, round, cast
    y = y / 255.0
    return (y - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
