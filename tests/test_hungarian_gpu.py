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


def test_fused_cost_matrix_is_the_torch_expression():
    """coda_matcher_cost_f32 against the reference's expressions (criterion.py:50-66, 1153): the L1 distances are
    torch.cdist's bit for bit, the gIoU is the stand-alone kernel's, the weighted cost is the torch arithmetic's."""
    from coda_neurips2023_amd import box_util
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    b, k1, k2, ncls = 5, 96, 17, 11
    def boxes(n):
        centre = torch.rand(b, n, 3, generator=gen) * 1.5
        size = torch.rand(b, n, 3, generator=gen) * 1.2 + 0.1
        angle = (torch.rand(b, n, generator=gen) - 0.5) * 6.0
        return box_util.get_3d_box_batch_tensor(size, angle, centre).to(dev)
    c1, c2 = boxes(k1), boxes(k2)
    a1, a2 = torch.rand(b, k1, 3, generator=gen).to(dev), torch.rand(b, k2, 3, generator=gen).to(dev)
    prob = torch.softmax(torch.randn(b, k1, ncls, generator=gen), -1).to(dev)
    obj = torch.rand(b, k1, generator=gen).to(dev)
    labels = torch.randint(0, ncls, (b, k2), generator=gen).to(dev)
    nums = torch.tensor([17, 3, 0, 9, 1]).to(dev)
    w = (1.0, 5.0, 5.0, 3.0)  # class, objectness, centre, gIoU (scripts/coda_sunrgbd_stage2.sh)
    for rotated in (True, torch.tensor(True, device=dev), False):
        gious, dist, cost = box_util.matcher_cost(c1, c2, nums, a1, a2, prob, labels, obj, w, rotated)
        ref_g = box_util.generalized_box3d_iou(c1, c2, nums, rotated_boxes=bool(rotated))
        ref_d = torch.cdist(a1, a2, p=1)
        class_mat = -torch.gather(prob, 2, labels.unsqueeze(1).expand(b, k1, k2))
        ref_c = w[0] * class_mat + w[1] * -obj.unsqueeze(-1) + w[2] * ref_d + w[3] * -ref_g
        assert torch.equal(gious, ref_g)
        assert torch.equal(dist, ref_d)
        assert torch.equal(cost, ref_c)


@pytest.mark.parametrize("rotated", [True, False])
def test_criterion_same_loss_with_and_without_the_fused_matcher_front(rotated):
    """SetCriterion.stacked_forward: fused cost + device solver + device scalars vs the reference-shaped route
    (torch expressions, scipy on the host, .item() scalars) -- same assignment, same loss, same gradients."""
    from types import SimpleNamespace

    from coda_neurips2023_amd import criterion as C
    from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig
    from coda_neurips2023_amd.box_util import get_3d_box_batch_tensor
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(21)
    nl, bsz, nq, ngt, ncls, nbin = 3, 4, 64, 16, 10, 12
    args = SimpleNamespace(**{a: 0 for a in C._WEIGHT_ARGS.values()})
    for k, v in dict(loss_no_object_weight=0.05, loss_angle_cls_weight=0.1, loss_angle_reg_weight=0.5,
                     loss_center_weight=5.0, loss_size_weight=1.0, loss_sem_cls_softmax_skip_none_gt_sample_weight=1,
                     matcher_giou_cost=3, matcher_cls_cost=1, matcher_center_cost=5, matcher_objectness_cost=5,
                     train_range_max=ncls).items():
        setattr(args, k, v)
    cfg = HotPathDatasetConfig(num_semcls=1, num_angle_bin=nbin)

    def rnd(*shape):
        return torch.rand(*shape, generator=gen)
    nactual = torch.tensor([5, 0, 16, 1])
    angles = (rnd(bsz, ngt) - 0.3) * 1.5 if rotated else torch.zeros(bsz, ngt)
    targets = {"gt_box_present": (torch.arange(ngt)[None] < nactual[:, None]).float(),
               "gt_box_sem_cls_label": torch.zeros(bsz, ngt, dtype=torch.int64),
               "gt_box_centers_normalized": rnd(bsz, ngt, 3), "gt_box_sizes_normalized": rnd(bsz, ngt, 3),
               "gt_box_angles": angles,
               "gt_box_corners": get_3d_box_batch_tensor(rnd(bsz, ngt, 3) + 0.2, angles, rnd(bsz, ngt, 3) * 3),
               "gt_angle_class_label": torch.randint(0, nbin, (bsz, ngt), generator=gen),
               "gt_angle_residual_label": (rnd(bsz, ngt) - 0.5) * 0.2}
    targets = {k: v.to(dev) for k, v in targets.items()}
    base = {"sem_cls_logits": torch.randn(nl, bsz, nq, 2, generator=gen), "angle_logits": torch.randn(nl, bsz, nq, nbin, generator=gen),
            "angle_residual_normalized": torch.randn(nl, bsz, nq, nbin, generator=gen) * 0.1,
            "center_normalized": rnd(nl, bsz, nq, 3), "size_normalized": rnd(nl, bsz, nq, 3)}
    corners = get_3d_box_batch_tensor(rnd(nl * bsz, nq, 3) + 0.2, (rnd(nl * bsz, nq) - 0.5) * 3, rnd(nl * bsz, nq, 3) * 3)
    results = []
    for fused in (True, False):
        crit = C.build_criterion(args, cfg).to(dev)
        crit.fused_matcher_cost = crit.device_scalars = fused
        crit.matcher.solver = "auto" if fused else "scipy"
        leaves = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        stacked = dict(leaves, box_corners=corners.view(nl, bsz, nq, 8, 3).to(dev))
        stacked["sem_cls_prob"] = torch.softmax(leaves["sem_cls_logits"], -1)[..., :-1]
        stacked["objectness_prob"] = 1 - torch.softmax(leaves["sem_cls_logits"], -1)[..., -1]
        outputs = {"outputs": {k: v[-1] for k, v in stacked.items()}, "stacked_outputs": stacked}
        loss, loss_dict = crit(outputs, dict(targets))
        loss.backward()
        results.append((loss.detach(), {k: v.grad.clone() for k, v in leaves.items()},
                        {k: v.detach() for k, v in loss_dict.items()}))
    (la, ga, da), (lb, gb, db) = results
    assert torch.allclose(la, lb, rtol=1e-6, atol=1e-6)
    assert set(da) == set(db)
    for k in da:
        assert torch.allclose(da[k], db[k], rtol=1e-5, atol=1e-6), k
    for k in ga:
        assert torch.allclose(ga[k], gb[k], rtol=1e-5, atol=1e-7), k


def test_non_finite_costs_terminate_and_poison_the_problem():
    """A diverged model hands the matcher NaN / inf costs.  scipy raises ValueError there (and the reference's run
    dies); the device solver must neither hang nor index out of range, and it marks the problem: the matched mask
    of a problem with a non-finite cost is NaN everywhere, so the step's loss is NaN and engine.py:155-157 stops
    the run.  Problems with finite costs in the same launch are solved as usual."""
    gen = torch.Generator().manual_seed(3)
    nprob, nq, ngt = 6, 128, 16
    outputs, cost = _outputs(nprob, nq, ngt, gen)
    c = outputs["center_dist"]
    c[0] = float("nan")
    c[1, :, 3] = float("inf")
    c[2, 5] = float("-inf")
    c[3, ::2] = float("nan")
    nactual = torch.tensor([16, 16, 16, 16, 7, 0])
    targets = {"gt_box_sem_cls_label": torch.zeros((nprob, ngt), dtype=torch.int64, device="cuda:0"),
               "nactual_gt": nactual.cuda()}
    got = Matcher(0, 0, 0, 1, solver="device")(outputs, targets)
    torch.cuda.synchronize()
    inds, mask = got["per_prop_gt_inds"].cpu().numpy(), got["proposal_matched_mask"].cpu().numpy()
    for b in range(4):
        assert np.isnan(mask[b]).all(), f"problem {b} holds non-finite costs: its mask must be poisoned"
        assert ((inds[b] >= 0) & (inds[b] < ngt)).all()
    for b in (4, 5):
        rows = np.nonzero(mask[b])[0]
        assert np.isfinite(mask[b]).all() and len(rows) == int(nactual[b])
        assert sorted(inds[b, rows].tolist()) == list(range(int(nactual[b])))
    # the untouched problem is still solved optimally
    r, col = linear_sum_assignment(cost[4, :, :7].numpy().astype(np.float64))
    rows = np.nonzero(mask[4])[0]
    assert np.isclose(cost[4].numpy()[rows, inds[4, rows]].sum(), cost[4].numpy()[r, col].sum(), atol=1e-6)
