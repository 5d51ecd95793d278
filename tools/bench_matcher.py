"""Matcher: device shortest-augmenting-path solver vs the reference's host route (D2H + scipy per scene), on the
problems of one training step (8 decoder layers x 8 scenes, 256 proposals, up to 64 GT boxes)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from coda_neurips2023_amd.criterion import Matcher  # noqa: E402

dev = torch.device("cuda:0")
for nq, ngt, nprob in ((256, 64, 64), (512, 64, 64), (256, 64, 8)):
    gen = torch.Generator().manual_seed(0)
    cost = (torch.rand((nprob, nq, ngt), generator=gen) * 4 - 1).to(dev)
    outputs = {"sem_cls_prob": torch.zeros((nprob, nq, 3), device=dev), "objectness_prob": torch.zeros((nprob, nq), device=dev),
               "center_dist": cost, "gious": torch.zeros((nprob, nq, ngt), device=dev)}
    for fill in ("full", "typical"):
        nact = torch.full((nprob,), ngt) if fill == "full" else torch.randint(1, 16, (nprob,), generator=gen)
        targets = {"gt_box_sem_cls_label": torch.zeros((nprob, ngt), dtype=torch.int64, device=dev), "nactual_gt": nact.to(dev)}
        for solver in ("device", "scipy"):
            m = Matcher(0, 0, 0, 1, solver=solver)
            m(outputs, targets)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m(outputs, targets)
            torch.cuda.synchronize()
            print(f"nq={nq} ngt={ngt} problems={nprob} gt={fill:8s} {solver:7s} {(time.perf_counter() - t0) * 100:8.3f} ms / call")
