"""Box corner geometry used by the box decoder.

Mirror of the two batched corner builders the model calls through the dataset
config (utils/box_util.py:346-358 ``rotz_tensor_batch``, :383-424
``get_3d_box_batch_tensor_xyz``, :427-490 ``flip_axis_to_camera_tensor`` /
``roty_batch_tensor`` / ``get_3d_box_batch_tensor``).  Corner ORDER and the
sign conventions are the reference's (the loss and the evaluation index them).

``generalized_box3d_iou`` (utils/box_util.py:861-875) is the matcher's gIoU cost: one HIP launch
for the whole (B, K1, K2) matrix (``include/coda_box_ops.h``) instead of the reference's host-side
triple loop.
"""
import torch

from . import _lib

# corner sign patterns, one row per corner: (x, y, z) multipliers of (l, w|h, h|w)/2
_SIGNS_XYZ = ((-1, 1, 1), (1, 1, 1), (1, -1, 1), (-1, -1, 1),
              (-1, 1, -1), (1, 1, -1), (1, -1, -1), (-1, -1, -1))       # box_util.py:404-412
_SIGNS_CAM = ((1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1),
              (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1))       # box_util.py:470-478


def rotz_tensor_batch(t):
    out = torch.zeros(tuple(t.shape) + (3, 3), dtype=t.dtype, device=t.device)
    c, s = torch.cos(t), torch.sin(t)
    out[..., 0, 0] = c
    out[..., 0, 1] = -s
    out[..., 1, 0] = s
    out[..., 1, 1] = c
    out[..., 2, 2] = 1
    return out


def roty_batch_tensor(t):
    out = torch.zeros(tuple(t.shape) + (3, 3), dtype=t.dtype, device=t.device)
    c, s = torch.cos(t), torch.sin(t)
    out[..., 0, 0] = c
    out[..., 0, 2] = s
    out[..., 1, 1] = 1
    out[..., 2, 0] = -s
    out[..., 2, 2] = c
    return out


def flip_axis_to_camera_tensor(pc):
    """depth (X right, Y forward, Z up) -> camera (X right, Y down, Z forward)."""
    # reference: clone, index-permute (x, z, y), negate the middle component; written
    # without list indexing (a host->device index copy that cannot be graph-captured)
    return torch.stack((pc[..., 0], -pc[..., 2], pc[..., 1]), dim=-1)


_SIGN_CACHE = {}


def _sign_tensor(signs, dtype, device):
    """Constant (8,3) sign pattern, created once per (pattern, dtype, device): no host->device
    copy in the steady state (keeps the step graph-capturable)."""
    key = (signs, dtype, str(device))
    if key not in _SIGN_CACHE:
        _SIGN_CACHE[key] = torch.tensor(signs, dtype=dtype, device=device)
    return _SIGN_CACHE[key]


def _corners(box_size, angle, center, signs, dims, rot_fn):
    flat = angle.ndim == 2
    if flat:
        assert box_size.ndim == 3 and center.ndim == 3
        bsize, nprop = box_size.shape[0], box_size.shape[1]
        box_size = box_size.reshape(-1, box_size.shape[-1])
        angle = angle.reshape(-1)
        center = center.reshape(-1, 3)
    rot = rot_fn(angle)
    half = torch.stack([box_size[..., d] for d in dims], -1) / 2  # (..., 3) in corner-axis order
    sign = _sign_tensor(signs, box_size.dtype, box_size.device)  # (8, 3)
    corners = half.unsqueeze(-2) * sign  # (..., 8, 3)
    corners = torch.matmul(corners, rot.transpose(-1, -2))
    corners = corners + center.unsqueeze(-2)
    if flat:
        corners = corners.reshape(bsize, nprop, 8, 3)
    return corners


def get_3d_box_batch_tensor_xyz(box_size, angle, center):
    """Corners in the depth (xyz) frame, rotation about z by -angle."""
    return _corners(box_size, angle, center, _SIGNS_XYZ, (0, 1, 2), lambda a: rotz_tensor_batch(-a))


def get_3d_box_batch_tensor(box_size, angle, center):
    """Corners in the camera frame (x: l, y: h, z: w), rotation about y."""
    return _corners(box_size, angle, center, _SIGNS_CAM, (0, 2, 1), roty_batch_tensor)


# Deployments of the reference that built utils/box_intersection.pyx only clip the first four GT
# columns of a scene when boxes are rotated (`K2 = rect2.shape[2]`, box_intersection.pyx:181); set
# this to 4 to reproduce that, leave it at -1 for the documented (TorchScript) behaviour.
ROTATED_K2_LIMIT = -1


def _vols_mode(flag):
    """0: gIoU; 1: intersection volumes as generalized_box3d_iou_tensor computes them (with its axis-aligned pre-test);
    2 ("exact"): intersection volumes of every pair, the evaluation's box3d_iou."""
    return 2 if flag == "exact" else int(bool(flag))


def generalized_box3d_iou(corners1, corners2, nums_k2, rotated_boxes=True, return_inter_vols_only=False,
                          needs_grad=False):
    """corners1 (B,K1,8,3), corners2 (B,K2,8,3), nums_k2 (B) -> (B,K1,K2) gIoU (utils/box_util.py:861-875).
    ``rotated_boxes``: a bool, or a one-element bool / uint8 tensor on the boxes' device (the kernel reads it
    there: no host read-back).  Forward only: the reference differentiates it only for loss_giou_weight > 0, which no CoDA recipe uses."""
    if needs_grad:
        raise NotImplementedError("generalized_box3d_iou: gradients (loss_giou_weight > 0) are outside the hot "
                                  "path of the CoDA recipes (scripts/*.sh pass --loss_giou_weight 0)")
    if not corners1.is_cuda:
        raise RuntimeError("CPU not supported")
    assert corners1.dim() == 4 and corners2.dim() == 4 and corners1.shape[2:] == (8, 3) == corners2.shape[2:]
    assert corners1.shape[0] == corners2.shape[0]
    b, k1, k2 = corners1.shape[0], corners1.shape[1], corners2.shape[1]
    c1 = corners1.detach().to(torch.float32).contiguous()
    c2 = corners2.detach().to(torch.float32).contiguous()
    nums = nums_k2.to(device=c1.device, dtype=torch.int32).contiguous() if nums_k2 is not None else None
    out = torch.empty((b, k1, k2), dtype=torch.float32, device=c1.device)
    with torch.cuda.device(c1.device):
        nums_ptr = nums.data_ptr() if nums is not None else None
        if torch.is_tensor(rotated_boxes) and rotated_boxes.is_cuda:
            flag = rotated_boxes.reshape(-1)[:1].to(device=c1.device, dtype=torch.uint8)
            st = _lib.load().coda_generalized_box3d_iou_devflag_f32(
                c1.data_ptr(), c2.data_ptr(), nums_ptr, out.data_ptr(), b, k1, k2, flag.data_ptr(),
                _vols_mode(return_inter_vols_only), int(ROTATED_K2_LIMIT), _lib.current_stream_handle())
        else:
            st = _lib.load().coda_generalized_box3d_iou_f32(c1.data_ptr(), c2.data_ptr(), nums_ptr,
                                                            out.data_ptr(), b, k1, k2, int(bool(rotated_boxes)),
                                                            _vols_mode(return_inter_vols_only), int(ROTATED_K2_LIMIT),
                                                            _lib.current_stream_handle())
    _lib.check(st, "generalized_box3d_iou")
    return out


def matcher_cost(corners1, corners2, nums_k2, center1, center2, cls_prob, labels, objectness, weights,
                 rotated_boxes=True):
    """The matcher's inputs in one pass (coda_matcher_cost_f32): gIoU (B,K1,K2) as ``generalized_box3d_iou``,
    the L1 centre distances ``torch.cdist(center1, center2, p=1)`` and the cost matrix of criterion.py:50-66,
    ``w_class * -cls_prob[..., labels] + w_objectness * -objectness + w_center * L1 + w_giou * -gIoU`` with
    ``weights = (w_class, w_objectness, w_center, w_giou)``.  No gradients (the matcher runs under no_grad and
    the centre loss has its own pass).  Returns (gious, center_dist, cost)."""
    if not corners1.is_cuda:
        raise RuntimeError("CPU not supported")
    b, k1, k2 = corners1.shape[0], corners1.shape[1], corners2.shape[1]
    f32 = dict(dtype=torch.float32, device=corners1.device)
    c1, c2 = corners1.detach().to(torch.float32).contiguous(), corners2.detach().to(torch.float32).contiguous()
    a1, a2 = center1.detach().to(torch.float32).contiguous(), center2.detach().to(torch.float32).contiguous()
    prob, obj = cls_prob.detach().to(torch.float32).contiguous(), objectness.detach().to(torch.float32).contiguous()
    lab = labels.to(torch.int64).contiguous()
    assert a1.shape == (b, k1, 3) and a2.shape == (b, k2, 3) and prob.shape[:2] == (b, k1) and obj.shape == (b, k1)
    assert lab.shape == (b, k2)
    nums = nums_k2.to(device=c1.device, dtype=torch.int32).contiguous() if nums_k2 is not None else None
    gious, dist, cost = (torch.empty((b, k1, k2), **f32) for _ in range(3))
    with torch.cuda.device(c1.device):
        flag = None
        if torch.is_tensor(rotated_boxes) and rotated_boxes.is_cuda:
            flag = rotated_boxes.reshape(-1)[:1].to(device=c1.device, dtype=torch.uint8)
        st = _lib.load().coda_matcher_cost_f32(
            c1.data_ptr(), c2.data_ptr(), nums.data_ptr() if nums is not None else None, a1.data_ptr(), a2.data_ptr(),
            prob.data_ptr(), lab.data_ptr(), obj.data_ptr(), *(float(w) for w in weights), gious.data_ptr(),
            dist.data_ptr(), cost.data_ptr(), b, k1, k2, prob.shape[2], int(flag is None and bool(rotated_boxes)),
            flag.data_ptr() if flag is not None else None, int(ROTATED_K2_LIMIT), _lib.current_stream_handle())
    _lib.check(st, "coda_matcher_cost_f32")
    return gious, dist, cost
