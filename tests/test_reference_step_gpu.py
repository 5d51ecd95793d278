"""The product against THE REFERENCE'S OWN MODULES on one whole training step at the sizes of BASELINE.json's configs
(configs[2], configs[3]'s recipe, and configs[4]'s shape in float32).

tests/golden/step_full_<case>.npz were written by tests/golden/make_golden.py::golden_step_full: 8 scenes x 20 000 points
through /root/reference's models/model_3detr.py (pre-encoder, 3 encoder + 8 decoder layers, heads; :1767-1794),
criterion.py's SetCriterion.forward (:1162-1216) and backward, float32 on the host with the C oracle behind
pointnet2._ext.  This test rebuilds the same seeded inputs and weights (tests/golden/step_inputs.py,
golden/weights.py), runs the step on the GPU through this package, and compares

* the pre-encoder's furthest-point-sampling indices: bit-exact;
* the total loss and each of its terms: 1e-3 relative (the north-star tolerance);
* the 64 (decoder layer, scene) Hungarian assignments: identical;
* strided samples of every head output: 1e-3 in the relative L2 norm (and the tensor's norm);
* every parameter gradient, through the fixture's 1024 strided samples and its norm: 2e-3 on the sampled relative L2
  error and on the norm.  Measured on MI355X: loss 7e-8, loss terms <= 9e-7, 64 of 64 assignments, gradients <= 3.9e-4
  (configs[2]) / <= 4.8e-4 (configs[3]), the largest in the set-abstraction MLP (1e6-row sums in another order).  The
  float64-judged 1e-3 bound on complete tensors is tests/test_full_step_gpu.py's; the reference itself cannot be run
  in float64 here in reasonable time, so this test holds twice the north-star figure on samples instead -- a pin of
  the WHOLE step to the reference's own code at full size (a wrong term, scale, sign, layer order or reduction shows
  up at 1e-1 .. 1).  The test point sits off the ReLU kinks of the query projection (golden/step_inputs.py:
  condition_query_projection; the two conditioned biases travel in the fixture), and the token-wise MLPs over the 1024
  tokens of the 128-query case get twice the bound (one flipped ReLU decision there is (1..2)e-3)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden import step_inputs as SI  # noqa: E402
from golden.weights import fill_deterministic  # noqa: E402

import bench  # noqa: E402
from coda_neurips2023_amd.criterion import build_criterion  # noqa: E402
from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig  # noqa: E402
from coda_neurips2023_amd.model_3detr import build_model  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "step_full_%s.npz")
GOLDEN_F64 = os.path.join(os.path.dirname(__file__), "golden", "step_full_%s_f64.npz")
LOSS_TOL, OUT_TOL, GRAD_TOL = 1e-3, 1e-3, 2e-3
WHOLE_TOL = 1e-3  # north_star's tolerance, on WHOLE gradient tensors against the reference's float64 run (below)


def sketch(g, name, k=128):
    """tests/golden/make_golden.py::sketch, restated (the generator cannot travel).  Count sketch of a flat float64 tensor: every element is added, with a seeded random sign, to one of k seeded
    random buckets.  For two tensors a, b:  sum_b (sketch(a) - sketch(b))_b^2  is an unbiased estimate of |a - b|^2
    (relative standard deviation ~ sqrt(2 / k) = 12.5 % on the square, 6 % on the norm) -- EVERY element takes part, in
    1 KB per tensor and O(n) work."""
    import zlib
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 77)
    bucket = rng.integers(0, k, g.size)
    sign = rng.integers(0, 2, g.size).astype(np.float64) * 2.0 - 1.0
    return np.bincount(bucket, weights=sign * g, minlength=k)


def compare_whole_tensors(model, case, tol=WHOLE_TOL, few_tokens=False):
    """Every parameter gradient, WHOLE, against the reference's own modules run in FLOAT64
    (tests/golden/step_full_<case>_f64.npz: make_golden.py::golden_step_full_f64 -- the reference's model_3detr.py +
    criterion.py on the host in double precision, 2.5 minutes on 8 cores; per tensor its norm and a 128-number
    count sketch, 1 KB): the estimated relative L2 error |g - g_ref| / |g_ref| of the whole tensor is
    held at north_star's 1e-3.  This is the float64-judged whole-tensor bound VERDICT r4 asked for next to the sampled
    2e-3 against the float32 run; the estimator's own 6 % noise is inside the margin (measured worst: see DESIGN.md)."""
    z = np.load(GOLDEN_F64 % case)
    params = dict(model.named_parameters())
    names = [k[7:] for k in z.files if k.startswith("sketch/")]
    assert set(names) == {n for n, p in params.items() if p.grad is not None}
    scale = max(float(z[f"norm/{n}"]) for n in names)
    report = []
    for n in names:
        ref_norm = float(z[f"norm/{n}"])
        g = params[n].grad.detach().double().cpu().numpy().reshape(-1)
        if ref_norm < 1e-6 * scale:  # true gradient zero (a bias in front of a batch-statistics norm)
            assert float(np.linalg.norm(g)) < 1e-3 * scale, n
            continue
        d = sketch(g, n) - z[f"sketch/{n}"]
        report.append((float(np.sqrt(np.sum(d * d))) / ref_norm, n, g.size))
    report.sort(reverse=True)
    for e, n, size in report[:6]:
        print(f"  whole-tensor grad {n:66s} ({size:7d} elements) est. rel L2 vs the float64 reference {e:.2e}")
    lim = lambda n: tol * (2.0 if few_tokens and n.startswith(("mlp_heads.", "query_projection.")) else 1.0)  # noqa: E731
    bad = [(e, n) for e, n, _ in report if not e < lim(n)]
    assert not bad, bad[:10]
    return report[0][0]


def run_product(dev, case, z):
    """The product's step on `dev` with the fixture's inputs -> (loss, loss_dict, captured, outputs, model)."""
    return build_product(dev, case, z)()


def build_product(dev, case, z):
    """-> step(): one forward + criterion + backward of the product on the fixture's inputs; every call starts from
    zeroed gradients and returns (loss, loss_dict, captured, outputs, model)."""
    batch, seam = SI.build(case)
    args = SI.recipe(case)
    cfg = HotPathDatasetConfig()
    seam_dev = {k: v.to(dev) for k, v in seam.items()}

    def provider(inputs, outputs, curr_epoch=-1):
        outputs["gt_text_correlation_embedding"] = seam_dev["img_emb"]
        outputs["gt_text_correlation_embedding_mask"] = seam_dev["mask"]
        outputs["weak_box_cate_label"] = seam_dev["weak_label"]
        outputs["weak_confidence_weight"] = seam_dev["weak_conf"]
        return outputs

    model, _ = build_model(args, cfg, text_features_fg_norm=seam_dev["text"], region_embedding_provider=provider)
    fill_deterministic(model, seed=SI.WEIGHT_SEED)
    with torch.no_grad():
        model.logit_scale.fill_(SI.LOGIT_SCALE_PARAM)
        # the fixture's test point sits off the ReLU kinks of the query projection (SI.condition_query_projection)
        lin = [m for m in model.query_projection.layers if isinstance(m, (torch.nn.Conv1d, torch.nn.Linear))]
        lin[0].bias.copy_(torch.from_numpy(z["cond/qp_bias0"]))
        lin[1].bias.copy_(torch.from_numpy(z["cond/qp_bias2"]))
    model.to(dev).train()
    crit = build_criterion(args, cfg).to(dev)
    if dev.type == "cpu":
        from oracle import cpu_port
        crit.giou_fn = cpu_port.generalized_box3d_iou
    captured = {"inds": [], "mask": []}
    solve = crit.matcher.solve

    def spy(final_cost, nactual_gt):
        res = solve(final_cost, nactual_gt)
        captured["inds"].append(res["per_prop_gt_inds"].detach().cpu())
        captured["mask"].append(res["proposal_matched_mask"].detach().cpu())
        return res

    crit.matcher.solve = spy
    model.pre_encoder.register_forward_hook(lambda m, i, o: captured.__setitem__("sa_inds", o[2].detach().cpu()))
    dbatch = {k: v.to(dev) for k, v in batch.items()}

    def step():
        captured["inds"], captured["mask"] = [], []
        model.zero_grad(set_to_none=True)
        pred = model(dbatch, curr_epoch=0)
        loss, loss_dict = crit(pred, dbatch)
        loss.backward()
        return loss, loss_dict, captured, pred["outputs"], model

    return step


def compare(loss, loss_dict, captured, outputs, model, z, grad_tol=GRAD_TOL, few_tokens=False):
    assert np.array_equal(captured["sa_inds"].numpy().astype(np.int32), z["sa_inds"]), "pre-encoder FPS indices"
    ref = float(z["loss"])
    rel = abs(float(loss) - ref) / abs(ref)
    print(f"loss {float(loss):.6f}  reference {ref:.6f}  rel {rel:.2e}")
    assert rel < LOSS_TOL
    keys = [str(k) for k in z["loss_keys"]]
    assert set(loss_dict) == set(keys), set(loss_dict) ^ set(keys)
    worst = 0.0
    for k, rv in zip(keys, z["loss_vals"]):
        gv = float(loss_dict[k])
        if k.startswith("loss_cardinality"):  # logged only; an arg-max within round-off of a tie moves it by 1/B
            assert abs(gv - rv) <= 2.0 / SI.B + 1e-6, (k, gv, rv)
            continue
        e = abs(gv - rv) / max(abs(rv), 1e-3 * abs(ref))
        worst = max(worst, e)
        assert e < LOSS_TOL, f"{k}: {gv} vs the reference's {rv}"
    inds = torch.cat([t.reshape(-1, t.shape[-1]) for t in captured["inds"]]).numpy()
    mask = torch.cat([t.reshape(-1, t.shape[-1]) for t in captured["mask"]]).numpy()
    r_pairs = np.where(z["assign_mask"] > 0, z["assign_inds"].astype(np.int64), -1)
    g_pairs = np.where(mask > 0, inds.astype(np.int64), -1)
    assert r_pairs.shape == g_pairs.shape
    same = int((r_pairs == g_pairs).all(axis=1).sum())
    print(f"worst loss-term rel err {worst:.2e}; identical assignments in {same} of {8 * SI.B} problems")
    assert same == 8 * SI.B
    for k in ["sem_cls_logits", "text_correlation_embedding", "center_normalized", "size_normalized", "angle_logits",
              "angle_residual", "box_corners"]:
        t = outputs[k].detach().double().cpu().reshape(-1)
        d = z[f"out/{k}"]
        idx = np.linspace(0, t.numel() - 1, SI.SAMPLES).astype(np.int64)
        err = float(np.linalg.norm(t[idx].numpy() - d[2:]) / np.linalg.norm(d[2:]))
        print(f"  output {k:28s} sampled L2 {err:.2e}  norm {abs(float(t.norm()) - d[1]) / d[1]:.2e}")
        assert err < OUT_TOL and abs(float(t.norm()) - d[1]) / d[1] < OUT_TOL, f"output {k}: {err:.2e}"
    params = dict(model.named_parameters())
    gkeys = [k[5:] for k in z.files if k.startswith("grad/")]  # ("cond/..." entries are inputs, not results)
    assert set(gkeys) == {n for n, p in params.items() if p.grad is not None}
    scale = max(float(z[f"grad/{n}"][1]) for n in gkeys)
    report = []
    for n in gkeys:
        d = z[f"grad/{n}"].astype(np.float64)
        g = params[n].grad.detach().double().cpu().reshape(-1)
        idx = np.linspace(0, g.numel() - 1, min(SI.SAMPLES, g.numel())).astype(np.int64)
        if d[1] < 1e-6 * scale:  # true gradient zero (a bias in front of a batch-statistics norm): round-off both sides
            assert float(g.norm()) < 1e-3 * scale, n
            continue
        es = float(np.linalg.norm(g[idx].numpy() - d[2:]) / max(np.linalg.norm(d[2:]), 1e-30))
        en = abs(float(g.norm()) - d[1]) / d[1]
        report.append((max(es, en), es, en, n))
    report.sort(reverse=True)
    for m, es, en, n in report[:8]:
        print(f"  grad {n:70s} sampled L2 {es:.2e}  norm {en:.2e}")
    # token-wise MLPs over 1024 tokens (heads / query projection at 128 queries): twice the bound -- one ReLU decision
    # within round-off of zero moves such a tensor by (1..2)e-3 between any two float32 evaluation orders
    # (tests/test_full_step_gpu.py states the reasoning)
    lim = lambda n: grad_tol * (2.0 if few_tokens and n.startswith(("mlp_heads.", "query_projection.")) else 1.0)  # noqa: E731
    bad = [(m, n) for m, es, en, n in report if not m < lim(n)]
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(SI.CASES))
def test_whole_step_equals_the_reference_modules_at_full_size(dev, case):
    z = np.load(GOLDEN % case)
    loss, loss_dict, captured, outputs, model = run_product(dev, case, z)
    torch.cuda.synchronize()
    compare(loss, loss_dict, captured, outputs, model, z, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)
    if os.path.exists(GOLDEN_F64 % case):
        compare_whole_tensors(model, case, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["configs4_fp32"])
def test_whole_step_repeated_in_one_process(dev, case):
    """The open item of round 4: this case failed ONCE (output not kept) and never again.  The case is the only one on
    the two-workgroup sampling kernel and the 512-query attention shapes, so: the same step CODA_STRESS_STEP_REPS times
    (default 200) in one process on one model -- the first run held against the reference's fixture as above, every
    later run held against the first ON THE DEVICE: sampling indices bit-equal, the 64 assignments identical, loss and
    every gradient tensor within 1e-5 (relative L2; the step's only run-to-run freedom is the order of a few fp64 /
    two-addend atomics).  Any failure leaves the failing quantity in gpurun_out/failures/ (tests/conftest.py)."""
    reps = int(os.environ.get("CODA_STRESS_STEP_REPS", "200"))
    z = np.load(GOLDEN % case)
    step = build_product(dev, case, z)
    loss, loss_dict, captured, outputs, model = step()
    torch.cuda.synchronize()
    compare(loss, loss_dict, captured, outputs, model, z, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)
    first = {"loss": loss.detach().double(), "inds": captured["sa_inds"].clone(),
             "assign": torch.cat([t.reshape(-1) for t in captured["inds"]]).clone(),
             "grads": {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}}
    norms = {n: g.double().norm().clamp_min(1e-30) for n, g in first["grads"].items()}
    scale = max(float(v) for v in norms.values())
    for rep in range(1, reps):
        loss_r, _, cap_r, _, model_r = step()   # same model, same inputs, gradients from zero
        torch.cuda.synchronize()
        assert torch.equal(cap_r["sa_inds"], first["inds"]), f"run {rep}: sampling indices changed"
        assert torch.equal(torch.cat([t.reshape(-1) for t in cap_r["inds"]]), first["assign"]), f"run {rep}: assignments"
        e = abs(float(loss_r.detach().double() - first["loss"]) / float(first["loss"]))
        assert e < 1e-6, f"run {rep}: loss moved by {e:.2e}"
        worst = (0.0, "")
        for n, p in model_r.named_parameters():
            if p.grad is None:
                continue
            d = float((p.grad.double() - first["grads"][n].double()).norm() / max(float(norms[n]), 1e-6 * scale))
            worst = max(worst, (d, n))
        assert worst[0] < 1e-5, f"run {rep}: gradient {worst[1]} moved by {worst[0]:.2e}"
        del loss_r


@pytest.mark.parametrize("case", list(SI.CASES))
def test_cpu_port_equals_the_reference_modules_at_full_size(case):
    """The full-size CHECKER of tests/test_full_step_gpu.py -- this package's host graph with the oracle's kernels in
    the seams (oracle/cpu_port.py) -- pinned to the reference's modules at the same size: float32 on both sides, same
    library kernels underneath.  Measured: loss equal to 8e-8, loss terms 8e-7, 64 of 64 assignments; gradients
    <= 6.2e-5 (configs[2]) and <= 2.2e-4 (configs[3]) in the sampled relative L2 norm, except the angle-class head of
    configs[3] at 1.6e-3 (norm 3e-5: a handful of entries behind a ReLU decision within round-off of zero -- the two
    sides order the head's GEMM differently).  Bound: 3e-3 on every gradient tensor.  ~30 s and ~3 GB per case."""
    from oracle import cpu_port
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with cpu_port.patched():
        res = run_product(torch.device("cpu"), case, np.load(GOLDEN % case))
    compare(*res, np.load(GOLDEN % case), grad_tol=3e-3, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)
    if os.path.exists(GOLDEN_F64 % case):
        # the checker (torch-CPU float32 layers + the C oracle) against the float64 truth: measured <= 5.4e-4 on whole
        # tensors (configs[3]; 2.6e-4 configs[2], 4.8e-4 configs[4]'s shape) -- held at 2e-3, the product at 1e-3 (above)
        compare_whole_tensors(res[4], case, tol=2e-3, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)
