"""Evaluation post-processing (parse_predictions) at the evaluate loop's size -- 8 scenes x 256 proposals x 20 000
points: the device path (coda_neurips2023_amd.ap_calculator) vs the numpy oracle on this box's host cores.
(The reference's own route -- a scipy Delaunay triangulation per proposal -- is timed in the build container, see
DESIGN.md section 8; it cannot travel to the GPU box.)"""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from coda_neurips2023_amd import ap_calculator as AP  # noqa: E402
from coda_neurips2023_amd.box_util import get_3d_box_batch_tensor  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402
from oracle import eval_oracle as EO  # noqa: E402

dev = torch.device("cuda:0")
b, k, n, ncls = 8, 256, 20000, 10
gen = torch.Generator().manual_seed(0)
pc, mn, mx = make_batch(b, n, seed=5)
pts = torch.from_numpy(pc)
pick = torch.randint(0, n, (b, k), generator=gen)
centres = torch.gather(pts, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) + (torch.rand(b, k, 3, generator=gen) - 0.5) * 0.6
sizes = torch.rand(b, k, 3, generator=gen) * 1.0 + 0.05
angles = (torch.rand(b, k, generator=gen) - 0.5) * 3.0
corners = get_3d_box_batch_tensor(sizes, angles, torch.stack((centres[..., 0], -centres[..., 2], centres[..., 1]), -1))
probs = torch.softmax(torch.randn(b, k, ncls, generator=gen), -1)
obj = torch.rand(b, k, generator=gen)
cfg = AP.get_ap_config_dict(dataset_config=SimpleNamespace(num_semcls=ncls))
d = [t.to(dev) for t in (corners, probs, obj, pts)]
for what in ("mask", "lists"):
    fn = (lambda: AP.prediction_mask(*d, cfg)) if what == "mask" else (lambda: AP.parse_predictions(*d, cfg))
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r = fn()
    torch.cuda.synchronize()
    print(f"device {what:5s}: {(time.perf_counter() - t0) * 100:8.3f} ms / batch")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record()
cnt = AP.box_point_counts(d[0], d[3])
ev[1].record()
AP.nms_keep_mask(d[0], d[2], d[1].argmax(-1), cnt >= 5, cfg)
ev[2].record()
torch.cuda.synchronize()
print(f"kernels: point counts {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us ({b * k * n / ev[0].elapsed_time(ev[1]) / 1e6:.1f} G tests/s), "
      f"nms {ev[1].elapsed_time(ev[2]) * 1e3:.1f} us")
t0 = time.perf_counter()
EO.parse_predictions(corners.numpy(), probs.numpy(), obj.numpy(), pc, cfg)
print(f"numpy oracle (vectorised half-space test): {(time.perf_counter() - t0) * 1e3:8.1f} ms / batch")
