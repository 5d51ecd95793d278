"""Box corner geometry used by the box decoder.

Mirror of the two batched corner builders the model calls through the dataset
config (utils/box_util.py:346-358 ``rotz_tensor_batch``, :383-424
``get_3d_box_batch_tensor_xyz``, :427-490 ``flip_axis_to_camera_tensor`` /
``roty_batch_tensor`` / ``get_3d_box_batch_tensor``).  Corner ORDER and the
sign conventions are the reference's (the loss and the evaluation index them).
"""
import torch

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
