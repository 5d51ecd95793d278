"""Seeded inputs of the full-size whole-step fixture (tests/golden/step_full.npz): the SAME scenes, targets and
region-embedding seam tensors tests/test_full_step_gpu.py uses for configs[2].  Shared by the generator (which feeds
them to the REFERENCE's modules) and the test (which feeds them to this package): the fixture carries outputs only."""
import math

import torch
import torch.nn.functional as F

B, NCLS = 8, 10
# the per-GPU shares of BASELINE.json's configs: points per scene, queries, decoder width, loss recipe
# (stage 1 = scripts/coda_sunrgbd_stage1.sh:7-27 at its own shape: d_dec 512, 128 queries, L1 alignment term only)
CASES = {"configs2": dict(npoints=20000, nq=256, dec_dim=256, stage=2),
         "configs3": dict(npoints=20000, nq=128, dec_dim=512, stage=1),
         # configs[4]'s shape (ScanNet-sized clouds, 512 queries) in the reference's own float32 arithmetic: the two-
         # workgroup sampling kernel, the 40 000-point ball query and the 512-query decoder shapes against the reference
         # (the bf16-MFMA attention mode of configs[4] itself is held against the bf16-rounding oracle elsewhere)
         "configs4_fp32": dict(npoints=40000, nq=512, dec_dim=256, stage=2)}
WEIGHT_SEED = 23
SAMPLES = 1024            # strided entries kept per tensor
LOGIT_SCALE = 100.0       # clip(exp(ln 100), max=100): models/model_3detr.py:1796 with a released CLIP checkpoint
LOGIT_SCALE_PARAM = math.log(100.0)


def recipe(case):
    """The argparse namespace of the case (bench.recipe_args: main.py's defaults + the stage's script), dropout 0."""
    import bench
    c = CASES[case]
    args = bench.recipe_args(c["nq"], dec_dim=c["dec_dim"], enc_dropout=0.0, dec_dropout=0.0, mlp_dropout=0.0)
    if c["stage"] == 1:
        args.loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi_weight = 0
    return args


def build(case):
    """-> (batch dict of CPU tensors incl. the ground-truth entries, seam dict)."""
    NQ, NPOINTS = CASES[case]["nq"], CASES[case]["npoints"]
    import bench
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    gen = torch.Generator().manual_seed(11)
    seam = {"text": F.normalize(torch.randn(NCLS, 512, generator=gen), dim=-1),
            "img_emb": F.normalize(torch.randn(B, NQ, 512, generator=gen), dim=-1),
            "mask": (torch.rand(B, NQ, 1, generator=gen) < 0.25).float(),
            "weak_label": torch.randint(0, NCLS, (B, NQ), generator=gen),
            "weak_conf": torch.rand(B, NQ, generator=gen) * (torch.rand(B, NQ, generator=gen) < 0.5)}
    pc, mn, mx = make_batch(B, NPOINTS, seed=555)
    batch = {"point_clouds": torch.from_numpy(pc), "point_cloud_dims_min": torch.from_numpy(mn),
             "point_cloud_dims_max": torch.from_numpy(mx)}
    batch.update(bench.synthetic_targets(batch, torch.Generator().manual_seed(2)))
    return batch, seam


def condition_query_projection(model, batch, nq, margin=2e-5, rounds=16):
    """Moves the two biases of ``model.query_projection`` (Linear + ReLU + Linear + ReLU on the 8 x nq query tokens) by
    a few 1e-5 so that NO pre-activation of the step's inputs lies within ``margin`` x rms of zero.

    Why: among the 2 x 524 288 pre-activations of this small MLP about one lies within fp32 round-off of zero; two
    float32 GEMMs of different summation order then take different ReLU branches for it, and because the gradient of
    these 1024-token column sums is a sum of largely cancelling terms, ONE flipped entry moves the four gradient
    tensors by 4e-3 .. 7e-3 in the relative L2 norm (tools/diag_query_proj.py; seen when the library's kernel choice
    for this GEMM changed).  That is a property of the test point, not of an implementation -- the same reasoning as
    the kink-margin scene search of tests/golden/make_golden.py::golden_model.  Works on the reference's module and on
    this package's (CPU, before the model is moved to the device); everything is evaluated in float64 with the C
    oracle's sampling.  Returns the two conditioned biases (float32)."""
    import copy
    from oracle import pointnet2_oracle as O
    ext = O.TorchExt()
    xyz = batch["point_clouds"][..., :3].contiguous().float()
    pick = lambda pts, idx: torch.gather(pts, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))  # noqa: E731
    enc_xyz = pick(xyz, ext.furthest_point_sampling(xyz, 2048))
    query_xyz = pick(enc_xyz, ext.furthest_point_sampling(enc_xyz, nq))
    pos = copy.deepcopy(model.pos_embedding).cpu().double()
    dims = [batch["point_cloud_dims_min"].double(), batch["point_cloud_dims_max"].double()]
    with torch.no_grad():
        x = pos(query_xyz.double(), input_range=dims).permute(2, 0, 1)  # (nq, B, C) tokens
        lin = [m for m in model.query_projection.layers if isinstance(m, (torch.nn.Conv1d, torch.nn.Linear))]
        assert len(lin) == 2
        w = [m.weight.detach().cpu().double().reshape(m.weight.shape[0], -1) for m in lin]
        b = [m.bias.detach().cpu().double().clone() for m in lin]
        for _ in range(rounds):
            z1 = x @ w[0].t() + b[0]
            z2 = torch.relu(z1) @ w[1].t() + b[1]
            moved = False
            for z, bias in ((z1, b[0]), (z2, b[1])):
                lim = margin * float(z.pow(2).mean().sqrt())
                near = z.abs() < lim
                if near.any():
                    # per channel: the smallest shift (in steps of the margin) that leaves none of its entries close
                    for c in torch.nonzero(near.any(0).any(0)).flatten().tolist():
                        col = z[..., c].reshape(-1)
                        for k in (3, -3, 5, -5, 7, -7, 9, -9, 13, -13, 21, -21):
                            if bool(((col + k * lim).abs() >= lim).all()):
                                bias[c] += k * lim
                                break
                        else:
                            raise RuntimeError("query projection: no small shift clears channel %d" % c)
                    moved = True
                    break  # layer 1 moved: layer 2's inputs changed, evaluate again
            if not moved:
                break
        else:
            raise RuntimeError("query projection: could not move every pre-activation away from zero")
        out = [t.float() for t in b]
        for m, t in zip(lin, out):
            m.bias.copy_(t.to(m.bias.device))
    return out
