"""Synthetic SUN-RGBD / ScanNet shaped point clouds (numpy, seeded).

No dataset is available in this environment, so the parity tests and
``bench.py`` feed the hot path with clouds that have the statistics that matter
to it: points lying on surfaces (room planes + furniture cuboids, so a r=0.2 m
ball holds tens of points, not the handful a uniform volume gives), depth
noise, and exact duplicate points when a scene is short (the reference loaders
sub-sample WITH replacement then, utils/pc_util.py:24-33, which is what makes
FPS ties real).  Camera-centred upright coordinates like SUN-RGBD
(datasets/sunrgbd_anonymous_aligned_image.py:307,770-771).
"""
import numpy as np


def _sample_rect(rng, n, origin, u, v):
    a = rng.random((n, 1), dtype=np.float32)
    b = rng.random((n, 1), dtype=np.float32)
    return origin[None, :] + a * u[None, :] + b * v[None, :]


def make_scene(n, seed, short_fraction=0.25):
    """One scene: (n, 3) float32.

    ``short_fraction`` of the seeds produce a raw cloud with fewer than ``n``
    points, which is then sub-sampled with replacement (duplicates).
    """
    rng = np.random.default_rng(seed)
    w, d = rng.uniform(3.0, 7.0, 2)
    h = rng.uniform(2.5, 3.0)
    x0, y0, z0 = -w / 2, 0.4, -1.2  # camera at the origin, looking along +y
    if rng.random() < short_fraction:
        n_raw = int(n * rng.uniform(0.55, 0.95))
    else:
        n_raw = int(n * rng.uniform(1.05, 1.5))

    f32 = np.float32
    ex = np.array([w, 0, 0], f32)
    ey = np.array([0, d, 0], f32)
    ez = np.array([0, 0, h], f32)
    o = np.array([x0, y0, z0], f32)
    planes = [  # (origin, u, v): floor, back wall, left, right, ceiling
        (o, ex, ey),
        (o + ey, ex, ez),
        (o, ey, ez),
        (o + ex, ey, ez),
        (o + ez, ex, ey),
    ]
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in planes])
    n_planes = int(0.6 * n_raw)
    counts = rng.multinomial(n_planes, areas / areas.sum())
    chunks = [_sample_rect(rng, c, po, u, v) for (po, u, v), c in zip(planes, counts)]

    n_boxes = int(rng.integers(5, 16))
    n_furn = n_raw - n_planes
    box_counts = rng.multinomial(n_furn, np.full(n_boxes, 1.0 / n_boxes))
    for c in box_counts:
        size = rng.uniform(0.3, 2.0, 3).astype(f32)
        size[2] = min(size[2], h * 0.9)
        yaw = rng.uniform(0, np.pi)
        rot = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]], f32)
        centre = np.array([rng.uniform(x0 + 0.5, x0 + w - 0.5), rng.uniform(y0 + 0.5, y0 + d - 0.5),
                           z0 + size[2] / 2], f32)
        # surface points: pick a face per point (top + 4 sides), uniform on the face
        u = rng.random((c, 3), dtype=np.float32) - 0.5
        face = rng.integers(0, 5, c)
        u[face == 0, 2] = 0.5
        u[face == 1, 0] = 0.5
        u[face == 2, 0] = -0.5
        u[face == 3, 1] = 0.5
        u[face == 4, 1] = -0.5
        chunks.append((u * size[None, :]) @ rot.T + centre[None, :])
    pts = np.concatenate(chunks, 0).astype(f32)
    pts += rng.normal(0.0, 0.005, pts.shape).astype(f32)
    pts = pts[rng.permutation(len(pts))]
    # utils/pc_util.py:24-33 random_sampling: with replacement only when short
    replace = len(pts) < n
    choice = rng.choice(len(pts), n, replace=replace)
    return np.ascontiguousarray(pts[choice], dtype=f32)


def make_batch(b, n, seed):
    """(b, n, 3) float32 plus per-scene min / max (b, 3) as the loaders provide."""
    pc = np.stack([make_scene(n, seed + 7919 * i) for i in range(b)], 0)
    return pc, pc.min(1), pc.max(1)
