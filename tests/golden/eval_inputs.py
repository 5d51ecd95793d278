"""Seeded detection / ground-truth lists for the mAP fixture (tests/golden/eval_det.npz), shared by the generator
(the REFERENCE's APCalculator + eval_det) and tests/test_eval_det.py (this package's).  Camera-frame corners built by
this package's box_util on CPU torch (deterministic), boxes on the floor of a 6 x 6 m room."""
import numpy as np
import torch

NCLS, NSCAN = 14, 7


def build(seed=9):
    from coda_neurips2023_amd import box_util
    rs = np.random.RandomState(seed)

    def corners(centre, size, heading):
        c = box_util.get_3d_box_batch_tensor(torch.from_numpy(size[None].astype(np.float32)),
                                             torch.from_numpy(heading[None].astype(np.float32)),
                                             torch.from_numpy(centre[None].astype(np.float32)))
        return c[0].numpy()

    batches = []   # list of (batch_pred_map_cls, batch_gt_map_cls)
    for scan in range(NSCAN):
        ngt = rs.randint(0, 7)
        gc = np.stack((rs.rand(ngt) * 6 - 3, rs.rand(ngt) * 0.5 - 1.0, rs.rand(ngt) * 5 + 1), -1)
        gs = rs.rand(ngt, 3) * 1.2 + 0.3
        gh = (rs.rand(ngt) - 0.5) * 3
        gcls = rs.randint(0, NCLS, ngt)
        gcor = corners(gc, gs, gh) if ngt else np.zeros((0, 8, 3), np.float32)
        gt = [(int(gcls[j]), gcor[j]) for j in range(ngt)]
        preds = []
        for j in range(ngt):                      # detections around every GT box: good, sloppy, duplicate, wrong class
            for rep in range(rs.randint(1, 4)):
                jit = rs.randn(3) * (0.05 if rep == 0 else 0.25)
                pc = corners((gc[j] + jit)[None], (gs[j] * (1 + rs.randn(3) * 0.1))[None], np.array([gh[j] + rs.randn() * 0.2]))[0]
                cls = int(gcls[j]) if rs.rand() < 0.8 else int(rs.randint(0, NCLS))
                preds.append((cls, pc, float(rs.rand())))
        for _ in range(rs.randint(0, 6)):         # false positives
            pc = corners(np.array([[rs.rand() * 6 - 3, -0.8, rs.rand() * 5 + 1]]), rs.rand(1, 3) + 0.3, (rs.rand(1) - 0.5) * 3)[0]
            preds.append((int(rs.randint(0, NCLS)), pc, float(rs.rand())))
        batches.append(([preds], [gt]))
    return batches
