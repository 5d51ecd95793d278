"""The device assignment solver (csrc/hungarian.hip) against scipy.optimize.linear_sum_assignment, which is what
the reference's Matcher calls per scene (criterion.py:68-79)."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from coda_neurips2023_amd.criterion import Matcher

pytestmark = pytest.mark.gpu


def _outputs(nprob, nq, ngt, gen, integer=False):
    dev = torch.device("cuda:0")
    if integer:  # small integer costs: many exact ties
        cost = torch.randint(0, 6, (nprob, nq, ngt), generator=gen).float()
    else:
        cost = torch.rand((nprob, nq, ngt), generator=gen) * 4 - 1
    outputs = {"sem_cls_prob": torch.zeros((nprob, nq, 3), device=dev),
               "objectness_prob": torch.zeros((nprob, nq), device=dev),
               "center_dist": cost.to(dev), "gious": torch.zeros((nprob, nq, ngt), device=dev)}
    return outputs, cost


@pytest.mark.parametrize("nq,ngt,integer", [(256, 64, False), (512, 64, False), (128, 5, False), (64, 64, False),
                                            (1024, 16, False), (256, 64, True), (96, 33, True)])
def test_matches_scipy(nq, ngt, integer):
    gen = torch.Generator().manual_seed(nq * 131 + ngt)
    nprob = 24
    outputs, cost = _outputs(nprob, nq, ngt, gen, integer)
    nactual = torch.randint(0, ngt + 1, (nprob,), generator=gen)
    nactual[0], nactual[1] = 0, ngt
    targets = {"gt_box_sem_cls_label": torch.zeros((nprob, ngt), dtype=torch.int64, device="cuda:0"),
               "nactual_gt": nactual.cuda()}
    got = Matcher(cost_class=0, cost_objectness=0, cost_giou=0, cost_center=1, solver="device")(outputs, targets)
    ref = Matcher(cost_class=0, cost_objectness=0, cost_giou=0, cost_center=1, solver="scipy")(outputs, targets)
    inds, mask = got["per_prop_gt_inds"].cpu().numpy(), got["proposal_matched_mask"].cpu().numpy()
    c = cost.numpy().astype(np.float64)
    for b in range(nprob):
        n = int(nactual[b])
        rows = np.nonzero(mask[b])[0]
        assert len(rows) == n
        assert sorted(inds[b, rows].tolist()) == list(range(n))       # every real GT exactly once
        assert (inds[b][mask[b] == 0] == 0).all()
        if n == 0:
            continue
        r, col = linear_sum_assignment(c[b, :, :n])
        assert np.isclose(c[b, rows, inds[b, rows]].sum(), c[b, r, col].sum(), rtol=0, atol=1e-9)  # same optimum
    if not integer:  # no exact ties: the optimum is unique, so the assignment itself is scipy's
        assert torch.equal(got["per_prop_gt_inds"], ref["per_prop_gt_inds"])
        assert torch.equal(got["proposal_matched_mask"], ref["proposal_matched_mask"])
        for b in range(nprob):
            a, r = got["assignments"][b], ref["assignments"][b]
            assert len(a) == len(r)
            if len(r):
                assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1])


def test_outside_limits_falls_back_to_host():
    gen = torch.Generator().manual_seed(5)
    outputs, cost = _outputs(2, 2048, 8, gen)
    targets = {"gt_box_sem_cls_label": torch.zeros((2, 8), dtype=torch.int64, device="cuda:0"),
               "nactual_gt": torch.tensor([8, 3]).cuda()}
    auto = Matcher(0, 0, 0, 1)(outputs, targets)
    ref = Matcher(0, 0, 0, 1, solver="scipy")(outputs, targets)
    assert torch.equal(auto["per_prop_gt_inds"], ref["per_prop_gt_inds"])
    with pytest.raises(RuntimeError):
        Matcher(0, 0, 0, 1, solver="device")(outputs, targets)
